"""Counterpart of the reference's config/app_registration.py:1-5: the registry singleton plus a
helper that registers this backend's retriever stacks under the framework's own mechanism."""
from ..framework.register import Register
from .modules import (HipLogitRerankerConfig, HipQwen3RerankerConfig, HipShardedFlatVectorStoreConfig,
                      MultiPathRetrieverConfig, VectorStoreRetrieverConfig)

registrator = Register()


def register_dense_retriever(config_path: str, app_name: str = "hip_dense_retriever") -> None:
    registrator.register(config_path, app_name, VectorStoreRetrieverConfig)


def register_sharded_vectorstore(config_path: str, app_name: str = "hip_sharded_vectorstore") -> None:
    """A row-sharded store: call it on every rank of a `torchrun` job (the store's searches are collective)."""
    registrator.register(config_path, app_name, HipShardedFlatVectorStoreConfig)


def register_multipath_retriever(config_path: str, app_name: str = "hip_multipath_retriever") -> None:
    registrator.register(config_path, app_name, MultiPathRetrieverConfig)


def register_reranker(config_path: str, app_name: str = "hip_reranker") -> None:
    registrator.register(config_path, app_name, HipLogitRerankerConfig)


def register_qwen3_reranker(config_path: str, app_name: str = "hip_qwen3_reranker") -> None:
    registrator.register(config_path, app_name, HipQwen3RerankerConfig)
