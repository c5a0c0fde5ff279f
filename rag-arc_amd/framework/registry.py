"""Registry side of the framework: module base class, singleton helper and the process-wide
Register (reference: framework/module.py:9-11, framework/singleton_decorator.py:1-6,
framework/register.py:7-26).

Contract kept from the reference (pinned by its framework tests, restated in tests/test_framework.py):
  * a module is whatever a config's build() returns and carries that config as `.config`;
  * Register() always yields the same object; register() lets FileNotFoundError escape (the file is
    opened outside the guard), reports unreadable JSON / invalid configs with print() without
    touching the registry, and overwrites an existing name; get_object() raises KeyError.
"""
import json
from abc import ABC
from dataclasses import dataclass
from typing import Any, Callable, Dict


@dataclass
class AbstractModule(ABC):
    config: Any  # the AbstractConfig that built this module


def singleton(cls) -> Callable[..., Any]:
    """Decorator: the first call constructs, every later call returns that same instance
    (constructor arguments of later calls are ignored)."""
    holder: Dict[type, Any] = {}

    def instance(*args, **kwargs):
        try:
            return holder[cls]
        except KeyError:
            holder[cls] = cls(*args, **kwargs)
            return holder[cls]

    instance.__wrapped__ = cls
    return instance


@singleton
class Register:
    def __init__(self):
        self.registrations: Dict[str, AbstractModule] = {}

    def register(self, config_path: str, app_name: str, config_type) -> None:
        with open(config_path, "r") as handle:
            try:
                payload = json.loads(handle.read())
                self.registrations[app_name] = config_type(**payload).build()
            except Exception as problem:  # noqa: BLE001 - report and carry on, like the reference
                print(f"Error registering {app_name}, the config file is not valid\n {problem}")

    def get_object(self, app_name: str) -> AbstractModule:
        return self.registrations[app_name]
