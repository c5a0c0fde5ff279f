"""Qwen3Reranker on the MI355X (reference: core/rerank/Reranker_Qwen3.py:6-75).

Same steps as the reference: format_instruction (:23-27) -> process_inputs (:29-39: tokenise the pairs with
truncation to max_length - |prefix| - |suffix|, wrap in the prefix / suffix token ids, LEFT pad) ->
compute_logits (:41-49: LM forward, last-position logits of "no" and "yes", log_softmax over the two, exp) ->
stable descending sort, optional top-k (:70-74).  The LM forward runs in rarc_lm_yes_no_logits
(csrc/decoder.hip: RMSNorm, RoPE, grouped-query causal attention, SwiGLU, MFMA GEMMs), the score -> order step in
rarc_rerank_order.  Weights are a HuggingFace Qwen3ForCausalLM state dict kept fp16 in HBM.

The tokeniser is a callable `tokenize(text) -> list[int]`; `from_tokenizer` takes the byte-level BPE tokeniser of
rag_arc_amd.core.rerank.bpe (over the checkpoint's vocab.json + merges.txt or tokenizer.json — Qwen's vocabulary does
not ship here, no network) and derives what the reference's constructor does: `prefix_ids` / `suffix_ids`, the token
ids of the chat-template prefix and suffix (:14-17), and `yes_id` / `no_id`, those of "yes" / "no" (:18-19).
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from ...hip import binding as B
from ..utils.data_model import Document
from .base import RerankerBase
from .hip_reranker import HipLogitReranker


class HipCausalLM:
    """Qwen3-style decoder weights in HBM + the last-position (no, yes) logits of left-padded token batches."""

    def __init__(self, state_dict: Dict[str, "np.ndarray"], num_attention_heads: int, num_key_value_heads: int,
                 head_dim: int, rms_norm_eps: float = 1e-6, rope_theta: float = 1e6, device: int = 0, fold_norms: bool = True):
        import torch

        if not torch.cuda.is_available():
            raise B.RarcError("no ROCm device visible: the HIP reranker has no CPU fallback")
        self.torch, self.lib = torch, B.load_library()
        self.device = torch.device("cuda", device)
        sd = state_dict

        def f16(v):
            if isinstance(v, torch.Tensor):
                return v.detach().to(self.device, torch.float32).half().contiguous()
            return torch.as_tensor(np.asarray(v), dtype=torch.float32).to(self.device).half().contiguous()

        self.embed = f16(sd["model.embed_tokens.weight"])
        self.lm_head = f16(sd["lm_head.weight"]) if "lm_head.weight" in sd else self.embed   # tied embeddings
        self.final_norm = f16(sd["model.norm.weight"])
        self.vocab, self.hidden = int(self.embed.shape[0]), int(self.embed.shape[1])
        self.n_q, self.n_kv, self.head_dim = int(num_attention_heads), int(num_key_value_heads), int(head_dim)
        self.layers: List[dict] = []
        i = 0
        while f"model.layers.{i}.self_attn.q_proj.weight" in sd:
            p = f"model.layers.{i}."
            self.layers.append(dict(
                in_norm=f16(sd[p + "input_layernorm.weight"]),
                qkv_w=torch.cat([f16(sd[p + f"self_attn.{n}_proj.weight"]) for n in ("q", "k", "v")]).contiguous(),
                q_norm=f16(sd[p + "self_attn.q_norm.weight"]), k_norm=f16(sd[p + "self_attn.k_norm.weight"]),
                o_w=f16(sd[p + "self_attn.o_proj.weight"]),
                post_norm=f16(sd[p + "post_attention_layernorm.weight"]),
                gate_up_w=self._interleave8(f16(sd[p + "mlp.gate_proj.weight"]), f16(sd[p + "mlp.up_proj.weight"])),
                down_w=f16(sd[p + "mlp.down_proj.weight"])))
            if fold_norms:
                # RMSNorm folded into its consumers (include/rarc.h, RarcLmLayer): W ⊙ γ over the input columns, product in
                # fp32, one rounding.  Large batches then read the residual stream directly and scale their output rows.
                lay = self.layers[-1]
                lay["qkv_w_folded"] = (lay["qkv_w"].float() * lay["in_norm"].float()[None, :]).half().contiguous()
                lay["gate_up_w_folded"] = (lay["gate_up_w"].float() * lay["post_norm"].float()[None, :]).half().contiguous()
            i += 1
        if not self.layers:
            raise B.RarcError("state dict holds no model.layers.* tensors")
        self.inter = int(self.layers[0]["down_w"].shape[1])
        qkv_w = (self.n_q + 2 * self.n_kv) * self.head_dim
        if tuple(self.layers[0]["qkv_w"].shape) != (qkv_w, self.hidden):
            raise B.RarcError("q/k/v projection shapes do not match the head configuration")
        if self.head_dim not in (64, 128) or self.hidden % 128 or self.inter % 128 or qkv_w % 128 or self.n_q % self.n_kv:
            raise B.RarcError(f"unsupported decoder shape: hidden={self.hidden} inter={self.inter} heads={self.n_q}/{self.n_kv} "
                              f"head_dim={self.head_dim}")
        self._zero = torch.zeros(max(self.hidden, 2 * self.inter, qkv_w), dtype=torch.float16, device=self.device)
        self._layer_tab = (B.LmLayer * len(self.layers))(*[
            B.LmLayer(**{k: v.data_ptr() for k, v in w.items()}) for w in self.layers])
        self._model = B.LmModel(self.hidden, len(self.layers), self.n_q, self.n_kv, self.head_dim, self.inter, self.vocab,
                                float(rms_norm_eps), float(rope_theta), self.embed.data_ptr(), self.lm_head.data_ptr(),
                                self.final_norm.data_ptr(), self._zero.data_ptr(), self._layer_tab)
        self._ws = None

    def _interleave8(self, gate, up):
        """[2*inter][hidden]: 8 gate_proj rows, then the 8 up_proj rows of the same features, and so on — the layout
        RarcLmLayer.gate_up_w asks for (include/rarc.h): SwiGLU becomes the epilogue of the gate/up GEMM."""
        inter, hidden = gate.shape
        if inter % 8 or tuple(up.shape) != (inter, hidden):
            raise B.RarcError(f"gate_proj / up_proj shapes {tuple(gate.shape)} / {tuple(up.shape)}: need equal shapes, rows % 8 == 0")
        return self.torch.stack([gate.view(inter // 8, 8, hidden), up.view(inter // 8, 8, hidden)], dim=1).reshape(
            2 * inter, hidden).contiguous()

    def yes_no_logits(self, input_ids, attention_mask, no_id: int, yes_id: int):
        """input_ids / attention_mask: [n][L] LEFT padded (host arrays).  Returns fp16 device tensor [n][2] = (no, yes)."""
        t = self.torch
        ids = np.asarray(input_ids, dtype=np.int32)
        mask = np.asarray(attention_mask).astype(bool)
        if ids.ndim != 2 or mask.shape != ids.shape or ids.shape[0] == 0:
            raise ValueError("input_ids and attention_mask must be [n][L]")
        n, L = ids.shape
        if not mask[:, -1].all() or (np.diff(mask.astype(np.int8), axis=1) < 0).any():
            raise ValueError("batches must be LEFT padded (the reference's tokenizer uses padding_side='left')")
        if ids[mask].min() < 0 or ids[mask].max() >= self.vocab or not (0 <= no_id < self.vocab and 0 <= yes_id < self.vocab):
            raise ValueError(f"token ids must lie in [0, {self.vocab})")
        start = (L - mask.sum(axis=1)).astype(np.int32)
        # tokens (GEMM rows) must be a multiple of 128: pad the length on the LEFT to a multiple of 32 (masked
        # positions; rotary embeddings are relative, so shifting every real token by the same amount changes nothing)
        # and the batch to a multiple of 4 with one-token sequences
        L32 = -(-L // 32) * 32
        if L32 != L:
            ids = np.concatenate([np.zeros((n, L32 - L), np.int32), ids], axis=1)
            start = start + (L32 - L)
            L = L32
        n_pad = -(-n // 4) * 4
        if n_pad != n:
            ids = np.concatenate([ids, np.zeros((n_pad - n, L), np.int32)])
            start = np.concatenate([start, np.full(n_pad - n, L - 1, np.int32)])
        with t.cuda.device(self.device):
            st = t.cuda.current_stream(self.device).cuda_stream
            d_ids = t.from_numpy(np.ascontiguousarray(np.where(ids < 0, 0, ids))).to(self.device)
            d_start = t.from_numpy(np.ascontiguousarray(start)).to(self.device)
            need = int(self.lib.rarc_lm_workspace_bytes(ctypes.addressof(self._model), n_pad * L))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(need, dtype=t.uint8, device=self.device)
            out = t.empty((n_pad, 2), dtype=t.float16, device=self.device)
            B.check(self.lib.rarc_lm_yes_no_logits(ctypes.addressof(self._model), d_ids.data_ptr(), d_start.data_ptr(), n_pad,
                                                   L, int(no_id), int(yes_id), self._ws.data_ptr(), self._ws.numel(),
                                                   out.data_ptr(), st), "rarc_lm_yes_no_logits")
            return out[:n]


    def prefix_kv_device(self, d_ids, d_start):
        """Run shared prompt PREFIXES once (include/rarc.h: rarc_lm_prefix_kv): int32 device tensors [n_prefix][P] (LEFT
        padded) and [n_prefix] (first real token), n_prefix * P a multiple of 128.  Returns a handle for
        yes_no_logits_device(..., prefix=handle, prefix_of=...): every layer's k | v rows of the prefix tokens."""
        t = self.torch
        if d_ids.dtype != t.int32 or d_start.dtype != t.int32 or d_ids.ndim != 2 or not d_ids.is_cuda:
            raise ValueError("prefix_kv_device takes int32 device tensors [n_prefix][P], [n_prefix]")
        n, P = d_ids.shape
        if n * P == 0 or (n * P) % 128 or d_start.shape != (n,):
            raise ValueError("n_prefix * P must be a positive multiple of 128")
        with t.cuda.device(self.device):
            need = int(self.lib.rarc_lm_workspace_bytes(ctypes.addressof(self._model), n * P))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(need, dtype=t.uint8, device=self.device)
            cache = t.empty(int(self.lib.rarc_lm_prefix_cache_bytes(ctypes.addressof(self._model), n * P)), dtype=t.uint8,
                            device=self.device)
            d_start = d_start.contiguous()
            B.check(self.lib.rarc_lm_prefix_kv(ctypes.addressof(self._model), d_ids.contiguous().data_ptr(), d_start.data_ptr(), n, P,
                                               self._ws.data_ptr(), self._ws.numel(), cache.data_ptr(), cache.numel(),
                                               t.cuda.current_stream(self.device).cuda_stream), "rarc_lm_prefix_kv")
        return {"cache": cache, "n": n, "P": P, "start": d_start}

    def yes_no_logits_device(self, d_ids, d_start, no_id: int, yes_id: int, out=None, prefix=None, prefix_of=None):
        """yes_no_logits() for batches that are already on the device: int32 tensors [n][L] (LEFT padded) and [n]
        (index of each sequence's first real token), n * L a multiple of 128.  Nothing is copied, read back or
        validated on the host (the kernels clamp token ids into the table).  Returns fp16 [n][2] = (no, yes).
        With `prefix` (from prefix_kv_device) and `prefix_of` (int32 [n]: the prefix each sequence continues, -1 none) the
        batch holds only the REMAINDERS of the prompts; the logits are those of prefix + remainder."""
        t = self.torch
        if d_ids.dtype != t.int32 or d_start.dtype != t.int32 or d_ids.ndim != 2 or not d_ids.is_cuda:
            raise ValueError("yes_no_logits_device takes int32 device tensors [n][L], [n]")
        n, L = d_ids.shape
        if n * L == 0 or (n * L) % 128 or d_start.shape != (n,):
            raise ValueError("n * L must be a positive multiple of 128")
        if (prefix is None) != (prefix_of is None):
            raise ValueError("prefix and prefix_of go together")
        with t.cuda.device(self.device):
            need = int(self.lib.rarc_lm_workspace_bytes(ctypes.addressof(self._model), n * L))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(need, dtype=t.uint8, device=self.device)
            if out is None:
                out = t.empty((n, 2), dtype=t.float16, device=self.device)
            st = t.cuda.current_stream(self.device).cuda_stream
            if prefix is None:
                B.check(self.lib.rarc_lm_yes_no_logits(ctypes.addressof(self._model), d_ids.contiguous().data_ptr(),
                                                       d_start.contiguous().data_ptr(), n, L, int(no_id), int(yes_id),
                                                       self._ws.data_ptr(), self._ws.numel(), out.data_ptr(), st),
                        "rarc_lm_yes_no_logits")
            else:
                if prefix_of.dtype != t.int32 or prefix_of.shape != (n,) or not prefix_of.is_cuda:
                    raise ValueError("prefix_of must be an int32 device tensor [n]")
                B.check(self.lib.rarc_lm_yes_no_logits_prefixed(
                    ctypes.addressof(self._model), d_ids.contiguous().data_ptr(), d_start.contiguous().data_ptr(), n, L,
                    prefix_of.contiguous().data_ptr(), prefix["cache"].data_ptr(), prefix["n"], prefix["P"],
                    prefix["start"].data_ptr(), int(no_id), int(yes_id), self._ws.data_ptr(), self._ws.numel(), out.data_ptr(), st),
                    "rarc_lm_yes_no_logits_prefixed")
            return out

    def yes_no_logits_shared_prefix(self, seqs, no_id: int, yes_id: int, min_prefix: int = 16):
        """(no, yes) logits of token-id SEQUENCES (lists, unpadded) that may share a common beginning — the prompts of one
        rerank() call.  When their longest common prefix is worth it, it is run once and only the remainders go through the
        LM per sequence; otherwise the plain left-padded batch.  Returns a host fp16 array [n][2]."""
        t = self.torch
        n = len(seqs)
        lcp = min(len(s) for s in seqs)
        first = seqs[0]
        for s in seqs[1:]:
            i = 0
            m = min(lcp, len(s))
            while i < m and s[i] == first[i]:
                i += 1
            lcp = i
        lcp = min(lcp, min(len(s) for s in seqs) - 1)          # every sequence keeps at least one token of its own
        if n < 2 or lcp < min_prefix:
            L = max(len(s) for s in seqs)
            ids = np.zeros((n, L), np.int32)
            mask = np.zeros((n, L), np.int8)
            for r, s in enumerate(seqs):
                ids[r, L - len(s):] = s
                mask[r, L - len(s):] = 1
            return self.yes_no_logits(ids, mask, no_id, yes_id).cpu().numpy()
        if max(max(s) for s in seqs) >= self.vocab or min(min(s) for s in seqs) < 0:
            raise ValueError(f"token ids must lie in [0, {self.vocab})")
        P = -(-lcp // 32) * 32                                  # one prefix sequence, left padded; 4 rows make 128 tokens
        pre = np.zeros((4, P), np.int32)
        pre[:, P - lcp:] = first[:lcp]
        pstart = np.full(4, P - lcp, np.int32)
        Ls = max(len(s) - lcp for s in seqs)
        Ls = -(-Ls // 32) * 32
        n_pad = -(-n // 4) * 4
        ids = np.zeros((n_pad, Ls), np.int32)
        start = np.full(n_pad, Ls - 1, np.int32)
        for r, s in enumerate(seqs):
            rest = s[lcp:]
            ids[r, Ls - len(rest):] = rest
            start[r] = Ls - len(rest)
        with t.cuda.device(self.device):
            handle = self.prefix_kv_device(t.from_numpy(pre).to(self.device), t.from_numpy(pstart).to(self.device))
            z = self.yes_no_logits_device(t.from_numpy(ids).to(self.device), t.from_numpy(start).to(self.device), no_id, yes_id,
                                          prefix=handle, prefix_of=t.zeros(n_pad, dtype=t.int32, device=self.device))
        return z[:n].cpu().numpy()


class HipQwen3Reranker(HipLogitReranker):
    """Drop-in for the reference's Qwen3Reranker: `rerank(query, documents, k=None, batch_size=8)`."""

    # the chat-template wrapper the reference encodes once at construction (Reranker_Qwen3.py:16-17)
    PREFIX = ("<|im_start|>system\nJudge whether the Document meets the requirements based on the Query and the Instruct "
              "provided. Note that the answer can only be \"yes\" or \"no\".<|im_end|>\n<|im_start|>user\n")
    SUFFIX = "<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n"

    def __init__(self, lm: HipCausalLM, tokenize: Callable[[str], Sequence[int]], yes_id: int, no_id: int,
                 prefix_ids: Optional[Sequence[int]] = None, suffix_ids: Optional[Sequence[int]] = None,
                 max_length: int = 4096, instruction: Optional[str] = None, pad_id: int = 0,
                 device: Optional[int] = None, share_prefix: bool = True):
        super().__init__(self._logits, instruction=instruction,
                         device=lm.device.index if device is None else device)
        self.share_prefix = bool(share_prefix)
        self.lm, self.tokenize = lm, tokenize
        self.yes_id, self.no_id, self.pad_id = int(yes_id), int(no_id), int(pad_id)   # token_true_id / token_false_id (:14-15)
        self.prefix, self.suffix = self.PREFIX, self.SUFFIX
        # prefix_tokens / suffix_tokens: given, or encoded with the same tokenizer as the reference does
        self.prefix_ids = list(tokenize(self.PREFIX)) if prefix_ids is None else list(prefix_ids)
        self.suffix_ids = list(tokenize(self.SUFFIX)) if suffix_ids is None else list(suffix_ids)
        self.max_length = int(max_length)

    @classmethod
    def from_tokenizer(cls, lm: HipCausalLM, tokenizer, max_length: int = 4096, instruction: Optional[str] = None,
                       pad_token: str = "<|endoftext|>", device: Optional[int] = None):
        """What the reference's constructor derives from its tokenizer (Reranker_Qwen3.py:14-19): the ids of "yes" / "no",
        the encoded prefix and suffix; `tokenizer` has encode(text) -> ids and convert_tokens_to_ids(token)
        (rag_arc_amd.core.rerank.bpe.ByteLevelBPETokenizer over the checkpoint's vocab.json + merges.txt or tokenizer.json)."""
        yes_id, no_id = tokenizer.convert_tokens_to_ids("yes"), tokenizer.convert_tokens_to_ids("no")
        if yes_id is None or no_id is None:
            raise ValueError("the tokenizer has no 'yes' / 'no' tokens")
        pad = tokenizer.convert_tokens_to_ids(pad_token)
        return cls(lm, tokenizer.encode, yes_id=yes_id, no_id=no_id, max_length=max_length, instruction=instruction,
                   pad_id=0 if pad is None else pad, device=device)

    def process_inputs(self, pairs: Sequence[str]):
        """Reranker_Qwen3.py:29-39: truncate each pair's tokens, wrap in prefix / suffix, left pad.  Returns
        (input_ids, attention_mask) as host arrays [n][L]."""
        room = self.max_length - len(self.prefix_ids) - len(self.suffix_ids)
        seqs = [self.prefix_ids + list(self.tokenize(p))[: max(room, 0)] + self.suffix_ids for p in pairs]
        L = max(len(s) for s in seqs)
        ids = np.full((len(seqs), L), self.pad_id, np.int32)
        mask = np.zeros((len(seqs), L), np.int8)
        for r, s in enumerate(seqs):
            ids[r, L - len(s):] = s
            mask[r, L - len(s):] = 1
        return ids, mask

    def token_sequences(self, pairs: Sequence[str]) -> List[List[int]]:
        """The unpadded token sequences process_inputs pads: prefix + truncated pair tokens + suffix."""
        room = self.max_length - len(self.prefix_ids) - len(self.suffix_ids)
        return [self.prefix_ids + list(self.tokenize(p))[: max(room, 0)] + self.suffix_ids for p in pairs]

    def _logits(self, query: str, contents: Sequence[str]):
        pairs = [self.format_instruction(self.instruction, query, c) for c in contents]
        if self.share_prefix:
            # the prompts of one call differ only from the document on: their common beginning runs through the LM once
            z = self.lm.yes_no_logits_shared_prefix(self.token_sequences(pairs), self.no_id, self.yes_id)
        else:
            ids, mask = self.process_inputs(pairs)
            z = self.lm.yes_no_logits(ids, mask, self.no_id, self.yes_id).cpu().numpy()   # fp16 [n][2]: 4 bytes per pair
        return z[:, 0], z[:, 1]

    def compute_logits(self, inputs) -> List[float]:
        """Reranker_Qwen3.py:41-49 for one left-padded batch `inputs = (input_ids, attention_mask)`: last-position logits
        of "no" / "yes" -> log_softmax over the two -> exp, as python floats of the fp16 values."""
        ids, mask = inputs
        z = self.lm.yes_no_logits(ids, mask, self.no_id, self.yes_id).cpu().numpy()
        scores, _ = self.score_order(z[:, 0], z[:, 1])
        return [float(v) for v in scores[0].cpu().numpy()]

    def compute_scores(self, pairs, instruction=None, **kwargs) -> List[float]:
        """Reranker_Qwen3.py:51-55: `pairs` = [(query, document text), ...] -> p_yes per pair."""
        texts = [self.format_instruction(instruction, q, d) for q, d in pairs]
        if not self.share_prefix:
            return self.compute_logits(self.process_inputs(texts))
        z = self.lm.yes_no_logits_shared_prefix(self.token_sequences(texts), self.no_id, self.yes_id)
        scores, _ = self.score_order(z[:, 0], z[:, 1])
        return [float(v) for v in scores[0].cpu().numpy()]

    def rerank(self, query: str, documents, k: int = None, batch_size: int = 8, **kwargs):
        """Reranker_Qwen3.py:57-75.  The reference scores the documents eight at a time (`batch_size`); a left-padded batch
        gives every prompt the logits it would get alone, so the grouping is free: with share_prefix all prompts of the
        call go through the LM together (in groups of `lm_batch`, default 128), their common beginning once per group."""
        if not documents:
            return []
        group = max(int(batch_size), int(kwargs.get("lm_batch", 128))) if self.share_prefix else int(batch_size)
        z_no, z_yes = [], []
        for s0 in range(0, len(documents), group):
            a, b = self.logit_fn(query, [d.content for d in documents[s0:s0 + group]])
            z_no.extend(list(a))
            z_yes.extend(list(b))
        _, perm = self.score_order(z_no, z_yes)
        return self.apply_order(documents, perm[0].tolist(), k)
