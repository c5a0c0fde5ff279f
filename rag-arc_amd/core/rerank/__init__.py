from .base import RerankerBase  # noqa: F401
from .hip_reranker import HipLogitReranker  # noqa: F401
