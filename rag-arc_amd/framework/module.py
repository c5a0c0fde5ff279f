"""Base of everything a config can build (reference: framework/module.py:9-11)."""
from abc import ABC
from dataclasses import dataclass
from typing import TYPE_CHECKING

if TYPE_CHECKING:
    from .config import AbstractConfig


@dataclass
class AbstractModule(ABC):
    config: "AbstractConfig"
