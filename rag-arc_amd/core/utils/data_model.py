"""The record every retriever, store and fusion method passes around
(reference: core/utils/data_model.py:4-9 — `content`, `metadata`, optional `id`)."""
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, Mapping, Optional


@dataclass
class Document:
    content: str
    metadata: Dict[str, Any] = field(default_factory=dict)
    id: Optional[str] = None

    def to_dict(self) -> Dict[str, Any]:
        """Plain-dict form (used by the shard-file docstore)."""
        return asdict(self)

    @classmethod
    def from_dict(cls, data: Mapping[str, Any]) -> "Document":
        return cls(content=data["content"], metadata=dict(data.get("metadata") or {}), id=data.get("id"))
