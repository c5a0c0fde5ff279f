from .base import RerankerBase  # noqa: F401
from .hip_reranker import HipLogitReranker  # noqa: F401
from .hip_qwen3 import HipCausalLM, HipQwen3Reranker  # noqa: F401
