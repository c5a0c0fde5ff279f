"""BERT-family embedding provider on the MI355X (fills the reference's empty
`encapsulation/embeddings/` slot; behaviour of HuggingFaceEmbeddings,
core/file_management/embeddings/huggingface.py:105-145: newline -> space, encode, CLS pooling,
optional normalisation, python float lists out).

The forward pass runs in the rarc_enc_* HIP kernels (csrc/encoder.hip); weights are taken from a
HuggingFace `BertModel` state dict (same tensor names), stored fp16 in HBM.  Tokenisation is a host
callable `tokenize(text) -> list[int]`; `wordpiece.WordPieceTokenizer` is the BERT tokeniser over a supplied
vocab.txt (no vocabulary ships with this repo: there is no network here).
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from ...hip import binding as B
from .base import Embeddings


def load_state_dict(path: str) -> Dict[str, "np.ndarray"]:
    """A BertModel state dict (HuggingFace tensor names, with or without the "bert." prefix) from .safetensors
    or .npz, as numpy arrays."""
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        return dict(load_file(path))
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as data:
            return {k: data[k] for k in data.files}
    raise ValueError(f"unsupported weight file (want .safetensors or .npz): {path}")


def checkpoint_defaults(weights_path: Optional[str], state_dict) -> Dict[str, object]:
    """What the checkpoint itself says about layer_norm_eps / pooling / output normalisation, for a provider configured
    without them.  `SentenceTransformer(model_name)` (huggingface.py:96-98) takes all three from the checkpoint's files:
    the transformer's config.json (`layer_norm_eps`), `1_Pooling/config.json` (`pooling_mode_*`) and the presence of a
    Normalize module in modules.json; where those files are not next to the weights the model FAMILY decides — BERT
    (bge): eps 1e-12, CLS pooling; MPNet (all-mpnet-base-v2, the reference's default model, huggingface.py:6): eps 1e-5,
    mean pooling, normalised output.  BERT's defaults on an MPNet checkpoint give embeddings that are silently not the
    reference's."""
    import json
    import os

    names = list(state_dict)
    mpnet = any(k.endswith("encoder.relative_attention_bias.weight") for k in names)
    out: Dict[str, object] = dict(model_type="mpnet" if mpnet else "bert", layer_norm_eps=1e-5 if mpnet else 1e-12,
                                  pooling="mean" if mpnet else "cls", force_normalize=bool(mpnet), source="model family")
    folder = os.path.dirname(os.path.abspath(weights_path)) if weights_path else None
    if not folder:
        return out

    def read(*parts):
        path = os.path.join(folder, *parts)
        if os.path.exists(path):
            with open(path, encoding="utf-8") as fh:
                return json.load(fh)
        return None

    cfg = read("config.json")
    if cfg and "layer_norm_eps" in cfg:
        out["layer_norm_eps"], out["source"] = float(cfg["layer_norm_eps"]), "checkpoint files"
        if "num_attention_heads" in cfg:
            out["num_heads"] = int(cfg["num_attention_heads"])
    pool = read("1_Pooling", "config.json")
    if pool:
        if pool.get("pooling_mode_cls_token"):
            out["pooling"] = "cls"
        elif pool.get("pooling_mode_mean_tokens"):
            out["pooling"] = "mean"
        else:
            raise ValueError(f"{folder}/1_Pooling/config.json asks for a pooling mode other than cls / mean")
        out["source"] = "checkpoint files"
    mods = read("modules.json")
    if mods is not None:
        out["force_normalize"] = any(str(m.get("type", "")).endswith("Normalize") for m in mods)
    return out


def _mpnet_bucket(delta, num_buckets=32, max_distance=128):
    """transformers' MPNetEncoder.relative_position_bucket for delta = key - query (float32 arithmetic as there): half the
    buckets per direction, exact below num_buckets / 4, logarithmic up to max_distance.  tests/test_mpnet_oracle.py pins the
    oracle's copy of this function to transformers; tests/test_gpu_mpnet.py pins this one to the oracle's."""
    n = -np.asarray(delta, dtype=np.int64)
    nb = num_buckets // 2
    ret = (n < 0).astype(np.int64) * nb
    n = np.abs(n)
    max_exact = nb // 2
    with np.errstate(divide="ignore"):
        val = np.log(n.astype(np.float32) / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact)) \
            * np.float32(nb - max_exact)
    val = np.where(np.isfinite(val), val, 0).astype(np.float32)
    return ret + np.where(n < max_exact, n, np.minimum(max_exact + val.astype(np.int64), nb - 1))


def _mpnet_rel_bias_table(rel_weight, span: int) -> "np.ndarray":
    """relative_attention_bias.weight [buckets][heads] -> fp32 [heads][2*span - 1], entry [h][key - query + span - 1]."""
    w = np.asarray(rel_weight, dtype=np.float32)
    return np.ascontiguousarray(w[_mpnet_bucket(np.arange(-(span - 1), span), num_buckets=w.shape[0])].T)


class HipBertEncoder:
    """token ids [n_seq][seq_len] (+ lengths) -> fp32 embeddings [n_seq][hidden] on the device.  Takes a BertModel state
    dict (bge, MiniLM, gte ...) or an MPNetModel one (all-mpnet-base-v2, the reference's default: relative-position
    attention bias, no token types) — told apart by their tensor names.

    precision = "fp32" (default): the reference's arithmetic class — `SentenceTransformer(...)` at
    huggingface.py:96-98 loads fp32 weights and `.encode` runs an fp32 forward.  Weights are kept as split fp16
    pairs (22 bits) + fp32 everything else, the GEMMs run as split-operand fp16 MFMA with fp32 accumulation
    (csrc/encoder_f32.hip); embeddings agree with an fp32 forward to fp32 rounding noise.
    precision = "fp16": the fast forward of csrc/encoder.hip (fp16 weights / activations, fp32 accumulation),
    a 1e-3-class approximation — for callers who would pass `model_kwargs={"torch_dtype": float16}`."""

    def __init__(self, state_dict: Dict[str, "np.ndarray"], num_heads: int, layer_norm_eps: float = 1e-12,
                 device: int = 0, pooling: str = "cls", precision: str = "fp32", query_path: bool = True):
        import torch

        if not torch.cuda.is_available():
            raise B.RarcError("no ROCm device visible: the HIP encoder has no CPU fallback")
        self.torch, self.lib = torch, B.load_library()
        self.device = torch.device("cuda", device)
        self.eps = float(layer_norm_eps)
        if pooling not in ("cls", "mean"):
            raise ValueError("pooling must be 'cls' (bge) or 'mean' (all-MiniLM / gte style)")
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' (the reference's) or 'fp16'")
        self.precision = precision
        self._pool_bit = 2 if pooling == "mean" else 0
        sd = {}
        for k, v in state_dict.items():     # BertModel / MPNetModel names, bare or under the usual wrappers' prefixes
            for pre in ("0.auto_model.", "bert.", "mpnet."):
                if k.startswith(pre):
                    k = k[len(pre):]
            sd[k] = v
        # MPNet family (the reference's default checkpoint, huggingface.py:6): q / k / v / o under attention.attn, no token
        # types, position ids from 2, one relative-position bias for all layers (oracle.mpnet_forward_f32 is the restatement)
        self.model_type = "mpnet" if "encoder.relative_attention_bias.weight" in sd else "bert"

        def f32(name):
            v = sd[name]
            if isinstance(v, torch.Tensor):          # already a tensor (any device): no numpy round trip
                return v.detach().to(self.device, torch.float32).contiguous()
            return torch.as_tensor(np.asarray(v), dtype=torch.float32).to(self.device).contiguous()

        f16 = lambda name: f32(name).half().contiguous()
        ld = f32 if precision == "fp32" else f16
        self.word = ld("embeddings.word_embeddings.weight")
        self.pos = ld("embeddings.position_embeddings.weight")
        if self.model_type == "mpnet":
            self.pos = self.pos[2:].contiguous()                       # real tokens sit at positions padding_idx + 1 + t
            self.type0 = torch.zeros(self.pos.shape[1], dtype=self.pos.dtype, device=self.device)
        else:
            self.type0 = ld("embeddings.token_type_embeddings.weight")[0].contiguous()
        self.emb_g, self.emb_b = ld("embeddings.LayerNorm.weight"), ld("embeddings.LayerNorm.bias")
        self.hidden = int(self.word.shape[1])
        self.heads = int(num_heads)
        if self.hidden % 128 or self.hidden // self.heads not in (32, 64) or self.hidden > 1024:
            raise B.RarcError(f"unsupported encoder shape: hidden={self.hidden}, heads={self.heads}")
        self.layers = []
        i = 0
        mp = self.model_type == "mpnet"
        qkv_names = ("attention.attn.q", "attention.attn.k", "attention.attn.v") if mp else \
            ("attention.self.query", "attention.self.key", "attention.self.value")
        o_name, ln1_name = ("attention.attn.o", "attention.LayerNorm") if mp else ("attention.output.dense", "attention.output.LayerNorm")
        while f"encoder.layer.{i}.{qkv_names[0]}.weight" in sd:
            p = f"encoder.layer.{i}."
            cat = lambda kind: torch.cat([ld(p + f"{n}.{kind}") for n in qkv_names]).contiguous()
            lay = dict(
                qkv_w=cat("weight"), qkv_b=cat("bias"),
                o_w=ld(p + o_name + ".weight"), o_b=ld(p + o_name + ".bias"),
                ln1_g=ld(p + ln1_name + ".weight"), ln1_b=ld(p + ln1_name + ".bias"),
                f1_w=ld(p + "intermediate.dense.weight"), f1_b=ld(p + "intermediate.dense.bias"),
                f2_w=ld(p + "output.dense.weight"), f2_b=ld(p + "output.dense.bias"),
                ln2_g=ld(p + "output.LayerNorm.weight"), ln2_b=ld(p + "output.LayerNorm.bias"))
            if precision == "fp32":
                # what the fused FFN1 epilogue sizes its output scale with (rarc.h, RarcEnc32Layer.f1_colmax):
                # c[k] = max_j |W1[j][k]|, c[hidden] = max_j |b1[j]|
                lay["f1_colmax"] = torch.cat([lay["f1_w"].abs().amax(dim=0), lay["f1_b"].abs().amax().reshape(1)]).float().contiguous()
                for nm in ("qkv", "o", "f1", "f2"):     # fp32 [n][k] -> split image fp16 [n][3k] + inverse row scales
                    lay[nm + "_w3"], lay[nm + "_rw"] = self._split_weight(lay.pop(nm + "_w"))
                    # ... and the QUERY PATH's image of the same split weight (fragment-major [hi | lo], 4 bytes per
                    # parameter): forwards of 32 / 64 / 128 tokens — a single embed_query — stream these instead of
                    # running the 128 x 128 tile kernels (csrc/encoder_f32.hip, "The QUERY PATH")
                    lay[nm + "_wq"] = self._pack_query_weight(lay[nm + "_w3"]) if query_path else None
            self.layers.append(lay)
            i += 1
        if not self.layers:
            raise B.RarcError("state dict holds no encoder.layer.* tensors")
        w1 = self.layers[0]["f1_w3" if precision == "fp32" else "f1_w"]
        self.inter = int(w1.shape[0])
        if self.inter % 128 or (precision == "fp32" and self.inter > 4096):
            raise B.RarcError("intermediate size must be a multiple of 128 (fp32 mode: at most 4096)")
        self.max_pos = int(self.pos.shape[0])
        self.vocab = int(self.word.shape[0])
        # host-side table of device pointers handed to rarc_enc_forward (tensors above keep the memory alive)
        LayerT, ModelT = (B.Enc32Layer, B.Enc32Model) if precision == "fp32" else (B.EncLayer, B.EncModel)
        self._layer_tab = (LayerT * len(self.layers))(*[
            LayerT(**{k: (None if v is None else v.data_ptr()) for k, v in w.items()}) for w in self.layers])
        self.query_path = bool(query_path) and precision == "fp32"
        self.rel_bias, rel_span = None, 0
        if mp:   # [heads][2*span - 1] fp32: entry [h][key - query + span - 1] (csrc: added to the scaled scores)
            rel_span = min(self.max_pos, 512)
            w = f32("encoder.relative_attention_bias.weight").cpu().numpy()
            if w.shape[1] != self.heads:
                raise B.RarcError(f"relative_attention_bias holds {w.shape[1]} heads, num_heads is {self.heads}")
            self.rel_bias = torch.from_numpy(_mpnet_rel_bias_table(w, rel_span)).to(self.device).contiguous()
        self._model = ModelT(self.hidden, self.heads, self.inter, len(self.layers), self.eps, self.word.data_ptr(),
                             self.pos.data_ptr(), self.type0.data_ptr(), self.emb_g.data_ptr(),
                             self.emb_b.data_ptr(), self._layer_tab, self.vocab, self.max_pos,
                             self.rel_bias.data_ptr() if mp else None, rel_span)
        self._fwd = self.lib.rarc_enc32_forward if precision == "fp32" else self.lib.rarc_enc_forward
        self._ws_bytes = self.lib.rarc_enc32_workspace_bytes if precision == "fp32" else self.lib.rarc_enc_workspace_bytes
        self._ws = None

    def _split_weight(self, w32):
        t = self.torch
        n, k = int(w32.shape[0]), int(w32.shape[1])
        with t.cuda.device(self.device):
            w3 = t.empty((n, 3 * k), dtype=t.float16, device=self.device)
            rw = t.empty(n, dtype=t.float32, device=self.device)
            B.check(self.lib.rarc_enc32_split_weight(w32.data_ptr(), n, k, w3.data_ptr(), rw.data_ptr(),
                                                     t.cuda.current_stream(self.device).cuda_stream), "rarc_enc32_split_weight")
            t.cuda.current_stream(self.device).synchronize()     # w32 is released when this returns
        return w3, rw

    def _pack_query_weight(self, w3):
        t = self.torch
        n, k = int(w3.shape[0]), int(w3.shape[1]) // 3
        with t.cuda.device(self.device):
            wq = t.empty(n * k * 2, dtype=t.float16, device=self.device)
            B.check(self.lib.rarc_enc32_pack_query_weight(w3.data_ptr(), n, k, wq.data_ptr(),
                                                          t.cuda.current_stream(self.device).cuda_stream), "rarc_enc32_pack_query_weight")
        return wq

    QUERY_PATH_TOKENS = (32, 64, 128)    # padded token counts the query path takes (RarcEnc32Layer.*_wq, include/rarc.h)

    def forward(self, input_ids, lengths=None, normalize: bool = True, non_blocking: bool = False):
        """`non_blocking`: the ids travel from PINNED host memory and the call returns as soon as the forward is enqueued
        (the default copies from pageable memory, which makes the host wait for everything queued before it)."""
        t = self.torch
        ids = np.asarray(input_ids, dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError("input_ids must be [n_seq][seq_len]")
        n_seq, L = ids.shape
        if L > self.max_pos or L > 512:
            raise ValueError(f"sequence length {L} exceeds the model limit")
        lens = np.full(n_seq, L, np.int32) if lengths is None else np.asarray(lengths, np.int32)
        if n_seq == 0:
            raise ValueError("empty batch")
        if lens.shape != (n_seq,) or lens.min() < 1 or lens.max() > L:
            raise ValueError(f"lengths must be [n_seq] values in [1, {L}]")
        if ids.min() < 0 or ids.max() >= self.vocab:
            raise ValueError(f"token ids must lie in [0, {self.vocab}) (got {int(ids.min())}..{int(ids.max())})")
        # GEMM rows (tokens) must be a multiple of 128: pad the sequence length to a multiple of 32 with masked
        # tokens (keys >= length are ignored, only position 0 is pooled) and the batch to a multiple of 4 — a
        # single 7-token query runs 4 x 32 tokens, not 128 sequences.  (Position table too short for the padded
        # length: pad the batch only.)
        L32 = -(-L // 32) * 32
        if L32 != L and L32 <= min(self.max_pos, 512):
            ids = np.concatenate([ids, np.zeros((n_seq, L32 - L), np.int32)], axis=1)
            L = L32
        step = 128 // math.gcd(L, 128)
        n_pad = -(-n_seq // step) * step
        if self.query_path and n_seq * L <= 128:
            # the query path: 32 / 64 / 128 padded tokens — one 7-token query runs 32 tokens, not 4 x 32
            fit = next((m for m in self.QUERY_PATH_TOKENS if m >= n_seq * L and m % L == 0), None)   # (none for 96 tokens)
            n_pad = fit // L if fit else n_pad
        if n_pad != n_seq:
            ids = np.concatenate([ids, np.zeros((n_pad - n_seq, L), np.int32)])
            lens = np.concatenate([lens, np.ones(n_pad - n_seq, np.int32)])
        M, H, I = n_pad * L, self.hidden, self.inter
        with t.cuda.device(self.device):
            st = t.cuda.current_stream(self.device).cuda_stream
            if non_blocking:   # (the pinned allocator keeps a block until the copy that reads it has completed)
                d_ids = t.from_numpy(np.ascontiguousarray(ids)).pin_memory().to(self.device, non_blocking=True)
                d_lens = t.from_numpy(np.ascontiguousarray(lens)).pin_memory().to(self.device, non_blocking=True)
            else:
                d_ids = t.from_numpy(np.ascontiguousarray(ids)).to(self.device)
                d_lens = t.from_numpy(np.ascontiguousarray(lens)).to(self.device)
            need = int(self._ws_bytes(H, I, M))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(need, dtype=t.uint8, device=self.device)
            out = t.empty((n_pad, H), dtype=t.float32, device=self.device)
            # one foreign call per forward: the layer loop runs inside librarc_hip.so
            B.check(self._fwd(ctypes.addressof(self._model), d_ids.data_ptr(), d_lens.data_ptr(), n_pad, L,
                                              (1 if normalize else 0) | self._pool_bit, self._ws.data_ptr(), self._ws.numel(),
                                              out.data_ptr(), st),
                    "rarc_enc32_forward" if self.precision == "fp32" else "rarc_enc_forward")
            return out[:n_seq]


    def forward_device(self, d_ids, d_lens, normalize: bool = True, out=None):
        """forward() for token ids that are already on the device: int32 tensors [n_seq][seq_len] and [n_seq]
        with n_seq * seq_len a multiple of 128.  Nothing is copied or read back, so nothing is validated on the
        host: ids and lengths are the caller's responsibility (the kernels clamp them into range).  Runs on the
        CURRENT stream; `out` (fp32 [n_seq][hidden], optional) lets a caller that pipelines forwards on a side stream
        own the result buffers instead of taking a fresh allocation per call."""
        t = self.torch
        if d_ids.dtype != t.int32 or d_lens.dtype != t.int32 or d_ids.ndim != 2 or not d_ids.is_cuda:
            raise ValueError("forward_device takes int32 device tensors [n_seq][seq_len], [n_seq]")
        n_seq, L = d_ids.shape
        M = n_seq * L
        if M == 0 or (M % 128 and not (self.query_path and M in self.QUERY_PATH_TOKENS)) or L > min(self.max_pos, 512) \
                or d_lens.shape != (n_seq,):
            raise ValueError("n_seq * seq_len must be a positive multiple of 128 (or 32 / 64 tokens with the query path) and "
                             "seq_len within the position table")
        with t.cuda.device(self.device):
            st = t.cuda.current_stream(self.device).cuda_stream
            need = int(self._ws_bytes(self.hidden, self.inter, M))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(need, dtype=t.uint8, device=self.device)
            if out is None:
                out = t.empty((n_seq, self.hidden), dtype=t.float32, device=self.device)
            elif out.shape != (n_seq, self.hidden) or out.dtype != t.float32 or not out.is_contiguous() or out.device != self.device:
                raise ValueError("out must be a contiguous fp32 [n_seq][hidden] tensor on the encoder's device")
            B.check(self._fwd(ctypes.addressof(self._model), d_ids.contiguous().data_ptr(),
                              d_lens.contiguous().data_ptr(), n_seq, L, (1 if normalize else 0) | self._pool_bit,
                              self._ws.data_ptr(), self._ws.numel(), out.data_ptr(), st),
                    "rarc_enc32_forward" if self.precision == "fp32" else "rarc_enc_forward")
            return out


class HipBertEmbeddings(Embeddings):
    """Embeddings provider: tokenizer callable + HipBertEncoder."""

    def __init__(self, encoder: HipBertEncoder, tokenize: Callable[[str], Sequence[int]], max_length: int = 512,
                 batch_size: Optional[int] = None, normalize_embeddings: bool = True, pad_id: int = 0,
                 prompts: Optional[Dict[str, str]] = None, default_prompt_name: Optional[str] = None,
                 prompt_name: Optional[str] = None, prompt: Optional[str] = None, max_batch_tokens: int = 131072,
                 tokenize_window: int = 4096, **kwargs):
        super().__init__(**kwargs)
        self.encoder, self.tokenize = encoder, tokenize
        self.max_length = min(max_length, encoder.max_pos)
        # sequences per encoder call.  sentence-transformers' default is 32 (`encode(batch_size=32)`), which runs this
        # encoder's GEMMs latency-bound (bge-large, 32 x 32 tokens: 3.76 ms, 256 x 32: 18.3 ms — a fifth of the rate);
        # None = fill each call up to `max_batch_tokens` padded tokens instead (results do not depend on the batching
        # beyond fp32 rounding, see embed_documents_device)
        self.batch_size = None if batch_size is None else max(1, int(batch_size))
        self.max_batch_tokens = max(128, int(max_batch_tokens))
        self.tokenize_window = max(1, int(tokenize_window))   # texts tokenised per host call (ahead of the GPU)
        self.normalize, self.pad_id = normalize_embeddings, pad_id
        # sentence-transformers prompts, as the reference passes them (huggingface.py:26-37: model_kwargs
        # 'prompts' / 'default_prompt_name', encode_kwargs 'prompt_name' / 'prompt'): a string put in front of
        # every text before tokenisation; `prompt` wins over `prompt_name`, which wins over the default
        self.prompts = dict(prompts or {})
        for name in (default_prompt_name, prompt_name):
            if name is not None and name not in self.prompts:
                raise ValueError(f"prompt name {name!r} not found in the configured prompts {sorted(self.prompts)}")
        self.prompt = prompt if prompt is not None else self.prompts.get(prompt_name if prompt_name is not None
                                                                          else default_prompt_name, None)
        self.last_stats: Dict[str, float] = {}

    # -- host side: texts -> padded id arrays ----------------------------------------------------------------------
    def _tokenize_many(self, texts: List[str]):
        """(ids int32 [n][L], lens int32 [n]) of a list of texts: one call into the library's multi-threaded
        tokeniser when the tokenizer offers `encode_batch` (wordpiece.WordPieceTokenizer does), else text by text."""
        if hasattr(self.tokenize, "encode_batch"):
            return self.tokenize.encode_batch(texts, max_length=self.max_length)
        rows = [list(self.tokenize(x))[: self.max_length] or [self.pad_id] for x in texts]
        lens = np.array([len(r) for r in rows], dtype=np.int32)
        ids = np.full((len(rows), int(lens.max()) if rows else 1), self.pad_id, dtype=np.int32)
        for r, row in enumerate(rows):
            ids[r, : len(row)] = row
        return ids, lens

    def _batches(self, lens: "np.ndarray"):
        """Cut a run of sequences (given in the order they will be embedded) into encoder calls: `batch_size` sequences
        each, or as many as fit `max_batch_tokens` once padded to the call's longest (a multiple of 32)."""
        n, s = len(lens), 0
        while s < n:
            if self.batch_size is not None:
                e = min(n, s + self.batch_size)
            else:
                e, longest = s, 0
                while e < n:
                    longest_new = max(longest, -(-int(lens[e]) // 32) * 32)
                    if e > s and (e + 1 - s) * longest_new > self.max_batch_tokens:
                        break
                    longest, e = longest_new, e + 1
            yield s, e
            s = e

    def embed_documents_device(self, texts: List[str]):
        """Same embeddings as embed_documents, left on the device as one fp32 tensor [n][hidden]
        (lets a device-resident store ingest them without the python-list round trip of
        VectorStore_Faiss.py:169-170).

        Pipeline: texts are sorted longest first (as sentence-transformers does), tokenised `tokenize_window` at a time by
        the library's host threads, cut into encoder calls by token budget, and every call is only ENQUEUED — ids go up
        from pinned memory without blocking — so the host tokenises window i + 1 while the GPU still runs the calls of
        window i.  An embedding does not depend on its batch mates except through fp32 rounding: the GEMM kernels pick
        tile shapes and split-K slabs by batch size, so the same text embedded alone or in a batch agrees to ~1e-6 in
        L2, not bit for bit (tests/test_gpu_ingest.py pins that)."""
        import time

        t = self.encoder.torch
        texts = [x.replace("\n", " ") for x in texts]
        if self.prompt:
            texts = [self.prompt + x for x in texts]
        n = len(texts)
        if n == 0:
            return t.empty((0, self.encoder.hidden), dtype=t.float32, device=self.encoder.device)
        order = sorted(range(n), key=lambda i: -len(texts[i]))   # longest first, like sentence-transformers
        parts, tok_s, calls, tokens, padded = [], 0.0, 0, 0, 0
        for w0 in range(0, n, self.tokenize_window):
            idx = order[w0: w0 + self.tokenize_window]
            t0 = time.perf_counter()
            ids, lens = self._tokenize_many([texts[i] for i in idx])
            tok_s += time.perf_counter() - t0
            for s, e in self._batches(lens):
                L = int(lens[s:e].max())
                parts.append(self.encoder.forward(ids[s:e, :L], lens[s:e], self.normalize, non_blocking=True))
                calls, tokens, padded = calls + 1, tokens + int(lens[s:e].sum()), padded + (e - s) * (-(-L // 32) * 32)
        cat = parts[0] if len(parts) == 1 else t.cat(parts)
        if n == 1:      # a single query (embed_query): nothing to put back in order — no permutation upload, no index_copy launch
            self.last_stats = dict(texts=1, encoder_calls=calls, tokens=tokens, padded_tokens=padded, tokenize_seconds=tok_s)
            return cat
        out = t.empty_like(cat)
        perm = t.from_numpy(np.asarray(order, dtype=np.int64)).pin_memory().to(cat.device, non_blocking=True)
        out.index_copy_(0, perm, cat)
        self.last_stats = dict(texts=n, encoder_calls=calls, tokens=tokens, padded_tokens=padded, tokenize_seconds=tok_s)
        return out

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_device(texts).cpu().numpy().tolist() if texts else []

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents([text])[0]

    # embed_query IS embed_documents([text])[0] here, as in the reference (huggingface.py:145): a batch of queries is
    # therefore one encoder call.  (The vector store's batched search and its coalescing front look for these.)
    def embed_queries(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents(list(texts))

    def embed_queries_device(self, texts: List[str]):
        return self.embed_documents_device(list(texts))
