from .base import Embeddings  # noqa: F401
from .table import TableEmbeddings  # noqa: F401
