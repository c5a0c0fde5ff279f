"""Tagged configs + modules that register this backend through the framework (the pattern of
framework/config.py + framework/module.py in the reference; pinned by its 21 framework tests).

JSON -> <Config>(**data).build() -> module graph, e.g.

    {"type": "multipath_retriever", "top_k_per_retriever": 50, "fusion": {"type": "rrf", "k": 60.0},
     "retrievers": [{"type": "vectorstore_retriever", "search_type": "similarity",
                     "vectorstore": {"type": "hip_flat_vectorstore", "metric": "cosine",
                                     "embedding": {"type": "table_embeddings", "path": "emb.npz"},
                                     "corpus_path": "corpus.npz"}}]}
"""
from dataclasses import dataclass, field
from typing import ClassVar, Annotated, Any, Dict, List, Literal, Optional, Union

from pydantic import Field

from ..framework.config import AbstractConfig
from ..framework.module import AbstractModule


@dataclass
class BuiltModule(AbstractModule):
    """A module that owns one built object (`impl`) and forwards attribute access to it."""
    impl: Any = field(default=None, repr=False)

    def __getattr__(self, name):
        impl = self.__dict__.get("impl")
        if impl is None:
            raise AttributeError(name)
        return getattr(impl, name)


class TableEmbeddingsConfig(AbstractConfig):
    type: Literal["table_embeddings"] = "table_embeddings"
    path: str
    device: Optional[int] = None      # keep the table in this GPU's HBM (query vectors are gathered there)

    def build(self) -> AbstractModule:
        from ..encapsulation.embeddings.table import TableEmbeddings

        emb = TableEmbeddings.from_npz(self.path)
        if self.device is not None:
            emb.to_device(self.device)
        return BuiltModule(config=self, impl=emb)


class HipBertEmbeddingsConfig(AbstractConfig):
    """The MI355X encoder as a registered embedding provider (the reference's HuggingFaceEmbeddings slot,
    core/file_management/embeddings/huggingface.py:85-98,116-126: texts in, python float lists out).
    `weights_path`: a BertModel or MPNetModel state dict (HuggingFace tensor names; the reference's default checkpoint,
    all-mpnet-base-v2, is MPNet with pooling = "mean") as .safetensors or .npz; `vocab_path`: the checkpoint's vocab.txt."""
    type: Literal["hip_bert_embeddings"] = "hip_bert_embeddings"
    weights_path: str
    vocab_path: str
    num_heads: int
    normalize_embeddings: bool = True     # bge models: True (encode_kwargs={"normalize_embeddings": True})
    # None = what the CHECKPOINT says, as SentenceTransformer(model_name) does (config.json, 1_Pooling/config.json,
    # modules.json next to the weights), else the model family's: BERT / bge = cls + 1e-12, MPNet (all-mpnet-base-v2, the
    # reference's default model) = mean + 1e-5 + normalised output
    pooling: Optional[Literal["cls", "mean"]] = None
    layer_norm_eps: Optional[float] = None
    do_lower_case: bool = True
    max_length: int = 512
    # sequences per encoder call; None = by token budget (max_batch_tokens), which keeps the GEMMs out of their
    # latency-bound small-batch regime (32 sequences x 32 tokens run at a fifth of the large-batch rate)
    batch_size: Optional[int] = None
    max_batch_tokens: int = 131072
    device: int = 0
    # "fp32": the reference's arithmetic (SentenceTransformer loads fp32, huggingface.py:96-98) — split-operand MFMA GEMMs,
    # fp32 everywhere else; "fp16": the faster 1e-3-class forward (what model_kwargs={"torch_dtype": float16} would ask for)
    precision: Literal["fp32", "fp16"] = "fp32"
    # sentence-transformers prompts (huggingface.py:26-37): model_kwargs 'prompts' / 'default_prompt_name',
    # encode_kwargs 'prompt_name' / 'prompt'
    prompts: Dict[str, str] = Field(default_factory=dict)
    default_prompt_name: Optional[str] = None
    prompt_name: Optional[str] = None
    prompt: Optional[str] = None

    def build(self) -> AbstractModule:
        from ..encapsulation.embeddings.hip_bert import (HipBertEmbeddings, HipBertEncoder, checkpoint_defaults,
                                                         load_state_dict)
        from ..encapsulation.embeddings.wordpiece import WordPieceTokenizer

        sd = load_state_dict(self.weights_path)
        auto = checkpoint_defaults(self.weights_path, sd)
        eps = self.layer_norm_eps if self.layer_norm_eps is not None else auto["layer_norm_eps"]
        pooling = self.pooling if self.pooling is not None else auto["pooling"]
        enc = HipBertEncoder(sd, num_heads=self.num_heads, layer_norm_eps=eps, device=self.device, pooling=pooling,
                             precision=self.precision)
        specials = dict(cls_token="<s>", sep_token="</s>", pad_token="<pad>", mask_token="<mask>") \
            if enc.model_type == "mpnet" else {}                       # MPNetTokenizer's specials over the same WordPiece
        tok = WordPieceTokenizer.from_file(self.vocab_path, do_lower_case=self.do_lower_case,
                                           max_length=min(self.max_length, enc.max_pos), **specials)
        return BuiltModule(config=self, impl=HipBertEmbeddings(enc, tok, max_length=self.max_length,
                                                               batch_size=self.batch_size,
                                                               max_batch_tokens=self.max_batch_tokens,
                                                               normalize_embeddings=self.normalize_embeddings or bool(auto["force_normalize"]),
                                                               pad_id=tok.pad, prompts=self.prompts,
                                                               default_prompt_name=self.default_prompt_name,
                                                               prompt_name=self.prompt_name, prompt=self.prompt))


EmbeddingsConfig = Annotated[Union[TableEmbeddingsConfig, HipBertEmbeddingsConfig], Field(discriminator="type")]


class HipFlatVectorStoreConfig(AbstractConfig):
    type: Literal["hip_flat_vectorstore"] = "hip_flat_vectorstore"
    embedding: EmbeddingsConfig
    metric: Literal["cosine", "ip"] = "cosine"
    normalize_L2: bool = False
    device: int = 0
    storage: Literal["f16", "f8", "f32"] = "f16"  # rows in HBM: fp16, fp8 e4m3fn + per-row scale, or fp32 (the reference's)
    corpus_path: Optional[str] = None  # .npz with `texts` (and optional `ids`) to ingest at build time
    # a folder written by save_local (`<index_name>.rarc` shard file(s) + `<index_name>.pkl`): the rows stream into HBM at
    # storage / PCIe rate instead of being embedded again (the reference's FaissVectorStore.load_local, VectorStore_Faiss.py:452-482)
    index_path: Optional[str] = None
    index_name: str = "index"
    coalesce: bool = True              # concurrent one-query callers share scans (hip_flat._QueryCoalescer)
    coalesce_window_us: float = 0.0
    # row storage: `capacity` rows are backed at build time; with `growable` the rows live in a virtual-memory arena that
    # grows IN PLACE up to `max_rows` (0 = what the device could hold) — add_texts never copies the rows already stored
    capacity: int = 0
    max_rows: int = 0
    growable: Optional[bool] = None    # None = the engine's default

    def build(self) -> AbstractModule:
        import numpy as np

        from ..encapsulation.database.vector_db.hip_flat import HipFlatVectorStore

        if self.index_path:
            store = HipFlatVectorStore.load_local(self.index_path, self.embedding.build().impl, self.index_name,
                                                  device=self.device, coalesce=self.coalesce,
                                                  coalesce_window_us=self.coalesce_window_us, capacity=self.capacity,
                                                  max_rows=self.max_rows, growable=self.growable)
            if (store.metric, store.storage) != (self.metric, self.storage):
                raise ValueError(f"{self.index_path}: saved as metric={store.metric} storage={store.storage}, "
                                 f"config says metric={self.metric} storage={self.storage}")
        else:
            store = HipFlatVectorStore(self.embedding.build().impl, metric=self.metric, normalize_L2=self.normalize_L2,
                                       device=self.device, storage=self.storage, coalesce=self.coalesce,
                                       coalesce_window_us=self.coalesce_window_us, capacity=self.capacity,
                                       max_rows=self.max_rows, growable=self.growable)
        if self.corpus_path:
            data = np.load(self.corpus_path, allow_pickle=False)
            ids = [str(i) for i in data["ids"]] if "ids" in data else None
            store.add_texts([str(t) for t in data["texts"]], ids=ids)
        return BuiltModule(config=self, impl=store)


class HipShardedFlatVectorStoreConfig(AbstractConfig):
    """HipFlatVectorStore row-sharded over the GPUs of a node (BASELINE configs 4 / 5): build it under
    `torchrun --nproc-per-node N` and every rank holds rows [r*ceil(n/N), (r+1)*ceil(n/N)) of the corpus in its GPU's HBM,
    answers a query with its local exact top-k, and ONE RCCL all-gather + merge gives every rank the global answer
    (rag_arc_amd/encapsulation/database/vector_db/hip_sharded.py).  Without WORLD_SIZE (or with WORLD_SIZE = 1) it is a
    one-shard store.  The reference's slot is the same (`framework/register.py:15-21`: any config with a `type` tag and
    build()); it has no distributed code of its own."""
    type: Literal["hip_sharded_flat_vectorstore"] = "hip_sharded_flat_vectorstore"
    embedding: EmbeddingsConfig
    metric: Literal["cosine", "ip"] = "cosine"
    normalize_L2: bool = False
    storage: Literal["f16", "f8", "f32"] = "f16"
    corpus_path: Optional[str] = None     # .npz with `texts` (and optional `ids`): EVERY rank reads it, each keeps its slice
    capacity: int = 0                     # rows PER RANK backed at build time / the arena's limit (see HipFlatVectorStoreConfig)
    max_rows: int = 0
    growable: Optional[bool] = None
    backend: Literal["nccl", "gloo"] = "nccl"   # nccl = RCCL over xGMI (the product); gloo: rehearsals on one device / CPU
    one_device: bool = False              # rehearsal: every rank on cuda:0 (needs backend gloo; RCCL refuses it)

    def build(self) -> AbstractModule:
        import os

        import numpy as np
        import torch
        import torch.distributed as dist

        from ..encapsulation.database.vector_db.hip_sharded import HipShardedFlatVectorStore

        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = 0 if self.one_device else int(os.environ.get("LOCAL_RANK", "0"))
        if world > 1 and not dist.is_initialized():
            if self.one_device and self.backend != "gloo":
                raise ValueError("one_device needs backend gloo (RCCL refuses two ranks on one device)")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            rank = int(os.environ.get("RANK", "0"))
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        store = HipShardedFlatVectorStore(self.embedding.build().impl, metric=self.metric, normalize_L2=self.normalize_L2,
                                          device=local_rank, storage=self.storage, capacity=self.capacity,
                                          max_rows=self.max_rows, growable=self.growable)
        if self.corpus_path:
            data = np.load(self.corpus_path, allow_pickle=False)
            ids = [str(i) for i in data["ids"]] if "ids" in data else None
            store.add_texts([str(t) for t in data["texts"]], ids=ids)
        return BuiltModule(config=self, impl=store)


class VectorStoreRetrieverConfig(AbstractConfig):
    type: Literal["vectorstore_retriever"] = "vectorstore_retriever"
    vectorstore: Annotated[Union[HipFlatVectorStoreConfig, HipShardedFlatVectorStoreConfig], Field(discriminator="type")]
    search_type: str = "similarity"
    search_kwargs: Dict[str, Any] = Field(default_factory=dict)

    def build(self) -> AbstractModule:
        from ..core.retrieval.dense import VectorStoreRetriever

        return BuiltModule(config=self, impl=VectorStoreRetriever(self.vectorstore.build().impl,
                                                                  search_type=self.search_type,
                                                                  search_kwargs=dict(self.search_kwargs)))


class HipLogitRerankerConfig(AbstractConfig):
    """The yes/no-logit reranker as a registered module (core/rerank/Reranker_Qwen3.py:6-75).  The score -> order
    step runs in the rarc_rerank_order kernel; the (no, yes) logits come from `logits_path`, an .npz table of
    logits computed offline for (query, document) pairs: `queries` [nq], `docs` [nd], `z_no` / `z_yes` [nq][nd]
    fp16 — unknown pairs raise KeyError, like TableEmbeddings."""
    type: Literal["hip_logit_reranker"] = "hip_logit_reranker"
    logits_path: str
    instruction: Optional[str] = None
    device: int = 0

    def build(self) -> AbstractModule:
        from ..core.rerank.hip_reranker import HipLogitReranker, TableLogits

        return BuiltModule(config=self, impl=HipLogitReranker(TableLogits.from_npz(self.logits_path),
                                                              instruction=self.instruction, device=self.device))


class HipQwen3RerankerConfig(AbstractConfig):
    """The reference's Qwen3Reranker (core/rerank/Reranker_Qwen3.py:6-75) with the LM forward on the MI355X
    (rarc_lm_yes_no_logits).  `weights_path`: a Qwen3ForCausalLM state dict (.safetensors / .npz, HuggingFace names);
    `tokenizer_path`: the checkpoint's tokenizer.json, or `vocab_path` + `merges_path`: its vocab.json and merges.txt
    (byte-level BPE, rag_arc_amd.core.rerank.bpe); the head geometry comes from the checkpoint's config.json
    (Qwen3-Reranker-0.6B: 16 / 8 heads of 128)."""
    type: Literal["hip_qwen3_reranker"] = "hip_qwen3_reranker"
    weights_path: str
    tokenizer_path: Optional[str] = None
    vocab_path: Optional[str] = None
    merges_path: Optional[str] = None
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_length: int = 4096
    instruction: Optional[str] = None
    device: int = 0

    def build(self) -> AbstractModule:
        from ..core.rerank.bpe import ByteLevelBPETokenizer
        from ..core.rerank.hip_qwen3 import HipCausalLM, HipQwen3Reranker
        from ..encapsulation.embeddings.hip_bert import load_state_dict

        if self.tokenizer_path:
            import json

            with open(self.tokenizer_path, encoding="utf-8") as fh:
                kind = json.load(fh).get("model", {}).get("type")
            if kind == "BPE":
                tok = ByteLevelBPETokenizer.from_tokenizer_json(self.tokenizer_path)
            else:                                   # another tokenizer model: the library AutoTokenizer itself loads it with
                from tokenizers import Tokenizer

                lib_tok = Tokenizer.from_file(self.tokenizer_path)

                class _LibraryTokenizer:
                    encode = staticmethod(lambda text: lib_tok.encode(text, add_special_tokens=False).ids)
                    convert_tokens_to_ids = staticmethod(lib_tok.token_to_id)

                tok = _LibraryTokenizer
        elif self.vocab_path and self.merges_path:
            import json
            import os

            # the added tokens: the checkpoint's own list (tokenizer_config.json: added_tokens_decoder) when it lies next to
            # vocab.json; otherwise Qwen2 / Qwen3's, but only if they fit THIS vocabulary — the first special id must follow
            # the last vocab.json id and none of the tokens may already be an ordinary token; anything else fails loudly
            special = None
            cfg_path = os.path.join(os.path.dirname(os.path.abspath(self.vocab_path)), "tokenizer_config.json")
            if os.path.exists(cfg_path):
                with open(cfg_path, encoding="utf-8") as fh:
                    dec = json.load(fh).get("added_tokens_decoder") or {}
                special = {v["content"]: int(i) for i, v in dec.items()} or None
            if special is None:
                with open(self.vocab_path, encoding="utf-8") as fh:
                    vocab = json.load(fh)
                clash = [t for t in self.SPECIAL_TOKENS if t in vocab]
                if max(vocab.values()) + 1 != self.FIRST_SPECIAL_ID or clash:
                    raise ValueError(f"hip_qwen3_reranker: {self.vocab_path} is not Qwen2/Qwen3's vocabulary (last id "
                                     f"{max(vocab.values())}, expected {self.FIRST_SPECIAL_ID - 1}; special tokens already "
                                     f"present: {clash[:3]}): give tokenizer_path (tokenizer.json) or put the checkpoint's "
                                     f"tokenizer_config.json next to vocab.json")
                special = {t: i for i, t in enumerate(self.SPECIAL_TOKENS, start=self.FIRST_SPECIAL_ID)}
            tok = ByteLevelBPETokenizer.from_files(self.vocab_path, self.merges_path, special)
        else:
            raise ValueError("hip_qwen3_reranker: give tokenizer_path, or vocab_path and merges_path")
        lm = HipCausalLM(load_state_dict(self.weights_path), self.num_attention_heads, self.num_key_value_heads,
                         self.head_dim, rms_norm_eps=self.rms_norm_eps, rope_theta=self.rope_theta, device=self.device)
        rr = HipQwen3Reranker.from_tokenizer(lm, tok, max_length=self.max_length, instruction=self.instruction)
        return BuiltModule(config=self, impl=rr)

    # vocab.json carries no added tokens; Qwen2 / Qwen3 checkpoints list these, in this order, from id 151643 on
    # (tokenizer_config.json: added_tokens_decoder) — tokenizer.json has them inside and needs none of this
    FIRST_SPECIAL_ID: ClassVar[int] = 151643
    SPECIAL_TOKENS: ClassVar[tuple] = (
        "<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|object_ref_start|>", "<|object_ref_end|>", "<|box_start|>",
        "<|box_end|>", "<|quad_start|>", "<|quad_end|>", "<|vision_start|>", "<|vision_end|>", "<|vision_pad|>",
        "<|image_pad|>", "<|video_pad|>", "<tool_call>", "</tool_call>", "<|fim_prefix|>", "<|fim_middle|>",
        "<|fim_suffix|>", "<|fim_pad|>", "<|repo_name|>", "<|file_sep|>", "<tool_response>", "</tool_response>",
        "<think>", "</think>")


RerankerConfig = Annotated[Union[HipLogitRerankerConfig, HipQwen3RerankerConfig], Field(discriminator="type")]


class RRFusionConfig(AbstractConfig):
    type: Literal["rrf"] = "rrf"
    k: float = 60.0
    device: int = 0

    def build(self) -> AbstractModule:
        from ..core.utils.fusion import HipRRFusion

        return BuiltModule(config=self, impl=HipRRFusion(k=self.k, device=self.device))


class MultiPathRetrieverConfig(AbstractConfig):
    type: Literal["multipath_retriever"] = "multipath_retriever"
    retrievers: List[Annotated[Union[VectorStoreRetrieverConfig], Field(discriminator="type")]]
    fusion: RRFusionConfig = Field(default_factory=RRFusionConfig)
    top_k_per_retriever: int = 50

    def build(self) -> AbstractModule:
        from ..core.retrieval.multipath import MultiPathRetriever

        return BuiltModule(config=self, impl=MultiPathRetriever([r.build().impl for r in self.retrievers],
                                                                fusion_method=self.fusion.build().impl,
                                                                top_k_per_retriever=self.top_k_per_retriever))
