"""The subset of the reference's ServerArgs (python/sglang/srt/server_args.py:46-265) that the
hot path reads, same field names."""
import dataclasses
from typing import Optional


@dataclasses.dataclass
class ServerArgs:
    model_path: str = ""
    dtype: str = "auto"
    device: str = "cuda"
    tp_size: int = 1
    quantization: Optional[str] = None          # None | "w8a8_fp8"
    attention_backend: str = "mi355"
    sampling_backend: str = "mi355"
    max_total_tokens: Optional[int] = None
    max_running_requests: Optional[int] = None
    context_length: Optional[int] = None
    mem_fraction_static: float = 0.85
    disable_cuda_graph: bool = False
    cuda_graph_max_bs: int = 64
    cuda_graph_bs: Optional[list] = None
    graph_capture_fallback: bool = False        # not in the reference: a failed graph capture -> eager on all ranks instead of raising
    random_seed: int = 0
    simulate_tp_shard: bool = False             # measurement aid: one TP rank, collectives stubbed
    chunked_prefill_size: Optional[int] = None   # None / -1: prompts are never split
    kv_cache_dtype: str = "auto"                  # "auto" = model dtype | "fp8_e4m3"
    # Not in the reference: how the multi-head KV pool lies in HBM.  "token_major" is the reference's
    # [slots, heads, d]; "head_major" stores [heads, slots, d] and hands out the permuted [slots, heads, d] view,
    # so a kv head's rows of consecutive slots are adjacent (what one decode workgroup streams); "auto" picks
    # head_major on the GPU for 16-bit models.
    kv_layout: str = "auto"
    enable_mixed_chunk: bool = False              # decode requests ride along prefill batches as 1-token extends
    schedule_conservativeness: float = 1.0        # scales the initial new-token estimate (0.7) of admission
    page_size: int = 1                            # KV slots per page (> 1: page-aligned allocation)
    # The reference's flag (server_args.py: disable_custom_all_reduce, default False).  The one-launch
    # collectives are built on GPU ranks and kept only if their start-up self-test agrees bit for bit
    # with the process group's own collectives on this machine (else RCCL serves every call).
    disable_custom_all_reduce: bool = False
    custom_all_reduce_self_test: bool = True
    max_prefill_tokens: int = 16384
    dist_backend: Optional[str] = None          # default: nccl (RCCL) on GPU, gloo on CPU
    dist_init_method: Optional[str] = None
