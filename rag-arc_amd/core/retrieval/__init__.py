from .base import BaseRetriever  # noqa: F401
from .dense import VectorStoreRetriever  # noqa: F401
from .multipath import MultiPathRetriever  # noqa: F401
