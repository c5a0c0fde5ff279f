from .base import VectorStore  # noqa: F401
from .hip_flat import HipFlatVectorStore  # noqa: F401
from .hip_sharded import HipShardedFlatVectorStore  # noqa: F401
