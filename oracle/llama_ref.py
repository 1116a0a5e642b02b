"""CPU restatement of the end-to-end hot path for a Llama model: batch plumbing, paged KV
cache, per-layer op sequence, logits, greedy sampling.  TEST INFRASTRUCTURE.

Follows: managers/schedule_batch.py:1106-1303 (extend: slots 1.. handed out in order, table
write), :1503-1573 (decode: seq_lens += 1, one new slot per request);
mem_cache/allocator.py:124-145 (slot 0 never allocated); mem_cache/memory_pool.py:240-255,
457-458 (cache rows); models/llama.py:94-98,180-191,245-267,308-354 (op order);
layers/logits_processor.py:254-270,453-455,490 (last-token gather, matmul in weight dtype,
.float()); layers/sampler.py:72-76 (argmax).  Attention = torch_native_backend.py:27-180.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import ops_ref


class OracleLlama:
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], dtype: torch.dtype,
                 max_total_tokens: int = 4096, max_reqs: int = 64, context_len: Optional[int] = None,
                 fp8_w8a8: bool = False, layers: Optional[int] = None):
        self.cfg = cfg
        self.dtype = dtype
        self.h = cfg["hidden_size"]
        self.hq, self.hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
        self.d = cfg.get("head_dim") or self.h // self.hq
        self.L = layers if layers is not None else cfg["num_hidden_layers"]
        self.eps = cfg["rms_norm_eps"]
        self.vocab = cfg["vocab_size"]
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        rs = cfg.get("rope_scaling")
        llama3 = rs if rs and rs.get("rope_type", rs.get("type")) == "llama3" else None
        # non-CUDA platforms keep the cache in the model dtype (rotary_embedding.py:98-101)
        self.cache = ops_ref.cos_sin_cache(self.d, cfg["max_position_embeddings"],
                                           cfg["rope_theta"], llama3, dtype)
        ctx = context_len or cfg["max_position_embeddings"]
        self.req_to_token = torch.zeros((max_reqs, ctx + 4), dtype=torch.int32)
        self.k_cache = [torch.zeros((max_total_tokens + 1, self.hkv, self.d), dtype=dtype) for _ in range(self.L)]
        self.v_cache = [torch.zeros((max_total_tokens + 1, self.hkv, self.d), dtype=dtype) for _ in range(self.L)]
        self.max_total_tokens = max_total_tokens
        self.fp8 = fp8_w8a8
        self.clear()
        self.qkv = []
        self.gate_up = []
        for i in range(self.L):
            p = f"model.layers.{i}."
            self.qkv.append(torch.cat([self.w[p + "self_attn.q_proj.weight"], self.w[p + "self_attn.k_proj.weight"],
                                       self.w[p + "self_attn.v_proj.weight"]], 0))
            self.gate_up.append(torch.cat([self.w[p + "mlp.gate_proj.weight"], self.w[p + "mlp.up_proj.weight"]], 0))
        if fp8_w8a8:
            self._q = {}

    # ---- pools ----
    def clear(self):
        self.next_slot = 1            # allocator hands out 1, 2, 3, ... (slot 0 = padding sink)
        self.next_req = 0
        self.req_pool_indices = None
        self.seq_lens = None

    def _alloc(self, n: int) -> torch.Tensor:
        assert self.next_slot + n - 1 <= self.max_total_tokens, "oracle KV pool exhausted"
        out = torch.arange(self.next_slot, self.next_slot + n, dtype=torch.int64)
        self.next_slot += n
        return out

    # ---- linear (unquant.py:110-122 / fp8_utils.py:676-749 non-CUDA branch) ----
    def _linear(self, x, w, key=None):
        if not self.fp8 or key is None:
            return F.linear(x, w)
        if key not in self._q:
            wf = w.float()
            s = wf.abs().amax(1, keepdim=True).clamp(min=1e-10) / ops_ref.FP8_MAX
            self._q[key] = ((wf * (1.0 / s)).clamp(-ops_ref.FP8_MAX, ops_ref.FP8_MAX).to(torch.float8_e4m3fn), s)
        wq, ws = self._q[key]
        return ops_ref.fp8_linear(x, wq, ws)

    # ---- model ----
    def _forward(self, input_ids, positions, out_cache_loc, attend):
        w = self.w
        h = F.embedding(input_ids, w["model.embed_tokens.weight"])
        residual = None
        for i in range(self.L):
            p = f"model.layers.{i}."
            if residual is None:
                residual = h
                h = ops_ref.rmsnorm(h, w[p + "input_layernorm.weight"], self.eps)
            else:
                h, residual = ops_ref.rmsnorm(h, w[p + "input_layernorm.weight"], self.eps, residual)
            qkv = self._linear(h, self.qkv[i], ("qkv", i))
            q, k, v = qkv.split([self.hq * self.d, self.hkv * self.d, self.hkv * self.d], dim=-1)
            q, k = ops_ref.apply_rope(positions, q, k, self.d, self.cache, True)
            self.k_cache[i][out_cache_loc] = k.view(-1, self.hkv, self.d)
            self.v_cache[i][out_cache_loc] = v.reshape(-1, self.hkv, self.d)
            a = attend(i, q.reshape(-1, self.hq, self.d)).reshape(-1, self.hq * self.d)
            h = self._linear(a, w[p + "self_attn.o_proj.weight"], ("o", i))
            h, residual = ops_ref.rmsnorm(h, w[p + "post_attention_layernorm.weight"], self.eps, residual)
            gu = self._linear(h, self.gate_up[i], ("gu", i))
            h = self._linear(ops_ref.silu_and_mul(gu), w[p + "mlp.down_proj.weight"], ("down", i))
        h, _ = ops_ref.rmsnorm(h, w["model.norm.weight"], self.eps, residual)
        return h

    def _logits(self, h):
        head = self.w.get("lm_head.weight", self.w["model.embed_tokens.weight"])
        return torch.matmul(h.to(head.dtype), head.T)[:, : self.vocab].float()

    @torch.no_grad()
    def extend(self, prompts: List[List[int]], prefix_lens: Optional[List[int]] = None):
        """Prefill `prompts` as new requests (or, with prefix_lens, extend requests whose first
        prefix_lens[i] tokens are already cached under the same request slots).
        Returns (logits [B, vocab] fp32, out_cache_loc)."""
        bs = len(prompts)
        if prefix_lens is None:
            prefix_lens = [0] * bs
            self.req_pool_indices = torch.arange(self.next_req, self.next_req + bs, dtype=torch.int64)
            self.next_req += bs
        seq_lens = torch.tensor([len(p) for p in prompts], dtype=torch.int64)
        pre = torch.tensor(prefix_lens, dtype=torch.int64)
        ext = seq_lens - pre
        input_ids = torch.tensor([t for p, n in zip(prompts, prefix_lens) for t in p[n:]], dtype=torch.int64)
        out_cache_loc = self._alloc(int(ext.sum()))
        ops_ref.write_req_to_token(self.req_to_token, self.req_pool_indices, pre, seq_lens, out_cache_loc)
        positions, _ = ops_ref.compute_position(pre, ext)
        self.seq_lens = seq_lens
        gqa = self.hq != self.hkv
        scaling = self.d ** -0.5

        def attend(layer, q):
            return ops_ref.extend_attention(q, self.k_cache[layer], self.v_cache[layer], self.req_to_token,
                                            self.req_pool_indices, seq_lens, pre, ext, scaling, gqa, True)

        h = self._forward(input_ids, positions, out_cache_loc, attend)
        last = torch.cumsum(ext, 0) - 1
        return self._logits(h[last]), out_cache_loc

    @torch.no_grad()
    def decode(self, next_ids: torch.Tensor):
        bs = next_ids.shape[0]
        locs = self.seq_lens.clone()
        self.seq_lens = self.seq_lens + 1
        out_cache_loc = self._alloc(bs)
        self.req_to_token[self.req_pool_indices, locs] = out_cache_loc.to(torch.int32)
        positions = ops_ref.clamp_position(self.seq_lens)
        gqa = self.hq != self.hkv
        scaling = self.d ** -0.5

        def attend(layer, q):
            return ops_ref.decode_attention(q, self.k_cache[layer], self.v_cache[layer], self.req_to_token,
                                            self.req_pool_indices, self.seq_lens, scaling, gqa)

        h = self._forward(next_ids.to(torch.int64), positions, out_cache_loc, attend)
        return self._logits(h), out_cache_loc

    def generate_greedy(self, prompts: List[List[int]], steps: int):
        """Returns (tokens [B, steps], logits list of [B, vocab])."""
        logits, _ = self.extend(prompts)
        all_logits, toks = [logits], [ops_ref.greedy(logits)]
        for _ in range(steps - 1):
            logits, _ = self.decode(toks[-1])
            all_logits.append(logits)
            toks.append(ops_ref.greedy(logits))
        return torch.stack(toks, 1), all_logits
