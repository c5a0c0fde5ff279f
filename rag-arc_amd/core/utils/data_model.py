"""Document record (reference: core/utils/data_model.py:4-9)."""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional


@dataclass
class Document:
    content: str
    metadata: Dict[str, Any] = field(default_factory=dict)
    id: Optional[str] = None
