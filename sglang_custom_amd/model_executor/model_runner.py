"""ModelRunner: owns the model, the KV pools, the attention backend and the HIP-graph runner,
and runs one forward pass (python/sglang/srt/model_executor/model_runner.py: ctor :154,
init_memory_pool :1054-1294, init_attention_backend :1305-1391, forward :1535-1600,
forward_decode/extend :1466-1499, sample :1616-1647)."""
import logging
from typing import Dict, Optional, Tuple

import torch

from ..configs.model_config import ModelConfig
from ..distributed import parallel_state
from ..layers.attention.mi355_backend import Mi355AttnBackend
from ..layers.logits_processor import LogitsProcessorOutput
from ..layers.quantization.w8a8_fp8 import W8A8Fp8LinearMethod
from ..layers.sampler import Sampler
from ..mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
from ..mem_cache.memory_pool import MHATokenToKVPool, MLATokenToKVPool, ReqToTokenPool
from ..models.llama import LlamaForCausalLM
from ..server_args import ServerArgs
from .forward_batch_info import ForwardBatch

logger = logging.getLogger(__name__)


class ModelRunner:
    def __init__(self, model_config: ModelConfig, server_args: ServerArgs, gpu_id: int = 0,
                 tp_rank: int = 0, tp_size: int = 1,
                 weights: Optional[Dict[str, torch.Tensor]] = None, attn_backend_cls=None):
        self.model_config = model_config
        self.server_args = server_args
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.gpu_id = gpu_id
        self.dtype = model_config.dtype
        if server_args.device == "cuda":
            torch.cuda.set_device(gpu_id)
            self.device = f"cuda:{gpu_id}"
        else:
            self.device = server_args.device
        self.init_torch_distributed()

        quant_factory = None
        if server_args.quantization == "w8a8_fp8":
            quant_factory = W8A8Fp8LinearMethod
        elif server_args.quantization is not None:
            raise ValueError(f"unsupported quantization {server_args.quantization!r}")
        model_cls = LlamaForCausalLM
        if getattr(model_config, "is_mla", False):
            # registry lookup by config.architectures in the reference (model_loader/utils.py); two families here
            from ..models.deepseek_v2 import DeepseekV2ForCausalLM

            model_cls = DeepseekV2ForCausalLM
        self.model = model_cls(model_config, dtype=self.dtype, device=self.device,
                               quant_method_factory=quant_factory)
        if weights is not None:
            self.model.load_weights(weights)
        else:
            self.model.init_random_weights(seed=server_args.random_seed)
        self.sampler = Sampler()
        # model_runner.py:287-290: the window of sliding-window models (None for Llama / DeepSeek)
        self.sliding_window_size = getattr(self.model, "get_attention_sliding_window_size", lambda: None)()
        self.init_memory_pool()
        self.attn_backend = (attn_backend_cls or Mi355AttnBackend)(self)
        self.graph_runner = None
        self.graph_capture_error = None
        if not server_args.disable_cuda_graph and str(self.device).startswith("cuda"):
            self.init_cuda_graphs()

    def init_cuda_graphs(self):
        """Capture the decode graphs (model_runner.py:1393-1420 of the reference).  The outcome is agreed on
        by all TP ranks: a capture that failed on ONE rank (or a collective of its warm-up runs that gave up
        waiting for that rank) makes every rank take the same road -- raise together, or, with
        server_args.graph_capture_fallback, run eager on this same runner (no second process group, the
        one-launch collectives dropped because a timed-out barrier leaves them unusable)."""
        from .hip_graph_runner import HipGraphRunner

        err = None
        try:
            self.graph_runner = HipGraphRunner(self)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
            self.graph_runner = None
        if not parallel_state.tp_any(err is not None):
            return
        self.graph_runner = None
        self.graph_capture_error = err or "graph capture failed on another rank"
        torch.cuda.synchronize()
        if parallel_state.get_custom_all_reduce() is not None:
            parallel_state.drop_custom_all_reduce("dropped after a failed graph capture")
        if not getattr(self.server_args, "graph_capture_fallback", False):
            raise RuntimeError(f"HIP-graph capture failed ({self.graph_capture_error}); "
                               "set disable_cuda_graph or graph_capture_fallback")
        logger.warning("HIP-graph capture failed (%s); running eager", self.graph_capture_error)

    # ---- set-up ----
    def init_torch_distributed(self):
        if self.tp_size == 1:
            parallel_state.init_distributed_environment(1, 0)
            return
        if getattr(self.server_args, "simulate_tp_shard", False):
            parallel_state.init_simulated_shard(self.tp_size, self.tp_rank)
            return
        backend = self.server_args.dist_backend or (
            "nccl" if str(self.device).startswith("cuda") else "gloo")
        parallel_state.init_distributed_environment(
            self.tp_size, self.tp_rank, backend, self.server_args.dist_init_method,
            use_custom_all_reduce=(not self.server_args.disable_custom_all_reduce
                                   and str(self.device).startswith("cuda")),
            local_device=torch.device(self.device).index or 0,
            custom_all_reduce_self_test=self.server_args.custom_all_reduce_self_test)

    def init_memory_pool(self):
        cfg, sa = self.model_config, self.server_args
        kv_heads = cfg.get_num_kv_heads(self.tp_size)
        kvd = getattr(sa, "kv_cache_dtype", "auto") or "auto"
        if kvd not in ("auto", "fp8_e4m3"):
            raise ValueError(f"kv_cache_dtype {kvd!r}: expected 'auto' or 'fp8_e4m3'")
        self.kv_cache_dtype = torch.float8_e4m3fn if kvd == "fp8_e4m3" else self.dtype
        mla = getattr(cfg, "is_mla", False)
        if mla:
            # one latent row per token and layer (model_runner.py:950-957)
            cell = (cfg.kv_lora_rank + cfg.qk_rope_head_dim) * cfg.num_hidden_layers * torch.finfo(self.kv_cache_dtype).bits // 8
        else:
            cell = 2 * kv_heads * cfg.head_dim * cfg.num_hidden_layers * torch.finfo(self.kv_cache_dtype).bits // 8
        if sa.max_total_tokens is not None:
            self.max_total_num_tokens = sa.max_total_tokens
        else:
            # reference rule: rest_memory * fraction / bytes-per-token (model_runner.py:959-970)
            free, _ = torch.cuda.mem_get_info(self.gpu_id)
            self.max_total_num_tokens = int(free * sa.mem_fraction_static) // cell
        max_reqs = sa.max_running_requests or min(
            max(self.max_total_num_tokens // cfg.context_len * 512, 2048), 4096)
        self.req_to_token_pool = ReqToTokenPool(max_reqs, cfg.context_len + 4, self.device)
        page = max(1, int(getattr(sa, "page_size", 1) or 1))
        # whole pages only (model_runner.py:1101-1106 rounds the same way)
        self.max_total_num_tokens = self.max_total_num_tokens // page * page
        if mla:
            # model_runner.py:1192-1206
            self.token_to_kv_pool = MLATokenToKVPool(
                self.max_total_num_tokens, page, self.kv_cache_dtype, cfg.kv_lora_rank, cfg.qk_rope_head_dim,
                cfg.num_hidden_layers, self.device)
        else:
            layout = getattr(sa, "kv_layout", "auto")
            if layout not in ("auto", "token_major", "head_major"):
                raise ValueError(f"kv_layout {layout!r}: expected auto, token_major or head_major")
            head_major = layout == "head_major" or (
                layout == "auto" and str(self.device).startswith("cuda")
                and self.dtype in (torch.bfloat16, torch.float16) and cfg.head_dim % 8 == 0)
            self.token_to_kv_pool = MHATokenToKVPool(
                self.max_total_num_tokens, page, self.kv_cache_dtype, kv_heads, cfg.head_dim,
                cfg.num_hidden_layers, self.device, head_major=head_major)
        if page == 1:
            self.token_to_kv_pool_allocator = TokenToKVPoolAllocator(
                self.max_total_num_tokens, self.dtype, self.device, self.token_to_kv_pool)
        else:
            self.token_to_kv_pool_allocator = PagedTokenToKVPoolAllocator(
                self.max_total_num_tokens, page, self.dtype, self.device, self.token_to_kv_pool)

    # ---- forward ----
    def forward_decode(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        self.attn_backend.init_forward_metadata(forward_batch)
        return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch)

    def forward_extend(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        self.attn_backend.init_forward_metadata(forward_batch)
        return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch)

    def forward(self, forward_batch: ForwardBatch) -> Tuple[LogitsProcessorOutput, bool]:
        """Returns (logits_output, ran_under_graph) like the reference's forward (:1535-1560)."""
        mode = forward_batch.forward_mode
        if mode.is_decode() and self.graph_runner is not None and self.graph_runner.can_run(forward_batch):
            return self.graph_runner.replay(forward_batch), True
        if mode.is_decode():
            return self.forward_decode(forward_batch), False
        if mode.is_extend():
            return self.forward_extend(forward_batch), False
        raise ValueError(f"Invalid forward mode: {mode}")

    def sample(self, logits_output: LogitsProcessorOutput, forward_batch: ForwardBatch) -> torch.Tensor:
        return self.sampler(logits_output, forward_batch.sampling_info, forward_batch.return_logprob)
