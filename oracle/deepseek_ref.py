"""CPU restatement of the end-to-end hot path for a DeepSeek-V2/V3 model whose layers are all dense:
the batch plumbing, logits and sampling of llama_ref.OracleLlama around the multi-head latent
attention of mla_ref.OracleMLA.  TEST INFRASTRUCTURE.

Follows models/deepseek_v2.py:1740-1968 (decoder layer, dense branch: input_layernorm -> self_attn ->
post_attention_layernorm -> DeepseekV2MLP :167-219), :1970-2062 (model), :969-1039 (which attention
form a batch takes with the Triton backend: multi-head for a prefill without prefix, absorbed
otherwise).  The attention module itself is pinned by the reference's outputs (tests/golden/mla.npz);
the ASSEMBLED model is not: the reference's CPU backends cannot run its multi-head prefill on a latent
pool, so no reference output of the whole stack exists here -- parity of the assembly is unpinned,
each part is pinned.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import ops_ref
from .llama_ref import OracleLlama
from .mla_ref import OracleMLA


class OracleDeepseek(OracleLlama):
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], dtype: torch.dtype, max_total_tokens: int = 4096,
                 max_reqs: int = 64, context_len: Optional[int] = None, fp8_w8a8: bool = False):
        self.cfg, self.dtype = cfg, dtype
        self.fp8 = fp8_w8a8        # every decoder-layer projection as an fp8 W8A8 linear (llama_ref._linear); the head stays 16-bit
        self._q = {}
        self.h, self.L = cfg["hidden_size"], cfg["num_hidden_layers"]
        self.eps, self.vocab = cfg["rms_norm_eps"], cfg["vocab_size"]
        self.w = {k: v.to(dtype) for k, v in weights.items()}
        self.max_total_tokens = max_total_tokens
        ctx = context_len or cfg["max_position_embeddings"]
        self.req_to_token = torch.zeros((max_reqs, ctx + 4), dtype=torch.int32)
        qr = cfg.get("q_lora_rank")
        self.attn = []
        for i in range(self.L):
            p = f"model.layers.{i}.self_attn."
            names = (["q_a_layernorm", "q_b_proj"] if qr is not None else ["q_proj", "kv_a_proj_with_mqa"]) + [
                "kv_a_layernorm", "kv_b_proj", "o_proj"]
            w = {n: weights[p + n + ".weight"] for n in names}
            if qr is not None:
                w["fused_qkv_a_proj_with_mqa"] = torch.cat(
                    [weights[p + "q_a_proj.weight"], weights[p + "kv_a_proj_with_mqa.weight"]], 0)
            m = OracleMLA(w, dtype, cfg["hidden_size"], cfg["num_attention_heads"], cfg["qk_nope_head_dim"],
                          cfg["qk_rope_head_dim"], cfg["v_head_dim"], cfg["kv_lora_rank"], qr, theta=cfg["rope_theta"],
                          eps=self.eps, rope_scaling=cfg.get("rope_scaling"),
                          max_position=cfg["max_position_embeddings"], slots=max_total_tokens, fp8_w8a8=fp8_w8a8)
            m.req_to_token = self.req_to_token     # one request table for every layer
            self.attn.append(m)
        self.gate_up = [torch.cat([self.w[f"model.layers.{i}.mlp.gate_proj.weight"],
                                   self.w[f"model.layers.{i}.mlp.up_proj.weight"]], 0) for i in range(self.L)]
        self.clear()

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
            h = attend(i, h)
            h, residual = ops_ref.rmsnorm(h, w[p + "post_attention_layernorm.weight"], self.eps, residual)
            h = self._linear(ops_ref.silu_and_mul(self._linear(h, self.gate_up[i], ("gate_up", i))),
                             w[p + "mlp.down_proj.weight"], ("down", i))
        h, _ = ops_ref.rmsnorm(h, w["model.norm.weight"], self.eps, residual)
        return h

    @torch.no_grad()
    def extend(self, prompts: List[List[int]], prefix_lens: Optional[List[int]] = None):
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
        rpi = self.req_pool_indices

        def attend(layer, h):
            m = self.attn[layer]
            if int(pre.sum()) > 0:
                return m.absorb("extend", positions, h, rpi, seq_lens, pre, ext, out_cache_loc)
            q, k, v = m.normal_prepare(positions, h, out_cache_loc)
            return m.normal_core(q, k, v, rpi, seq_lens, pre, ext, out_cache_loc)

        h = self._forward(input_ids, positions, out_cache_loc, attend)
        return self._logits(h[torch.cumsum(ext, 0) - 1]), out_cache_loc

    @torch.no_grad()
    def decode(self, next_ids: torch.Tensor):
        bs = next_ids.shape[0]
        locs = self.seq_lens.clone()
        self.seq_lens = self.seq_lens + 1
        out_cache_loc = self._alloc(bs)
        self.req_to_token[self.req_pool_indices, locs] = out_cache_loc.to(torch.int32)
        positions = ops_ref.clamp_position(self.seq_lens)

        def attend(layer, h):
            return self.attn[layer].absorb("decode", positions, h, self.req_pool_indices, self.seq_lens, None, None,
                                           out_cache_loc)

        h = self._forward(next_ids.to(torch.int64), positions, out_cache_loc, attend)
        return self._logits(h), out_cache_loc
