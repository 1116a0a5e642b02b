"""Drop-in check against the REFERENCE's own objects (build container only):
    python tests/golden/gen_dropin_golden.py

Builds the reference's ModelRunner on CPU (tiny Llama, fp32, its torch-native backend), runs a
prefill + greedy decode steps, then replaces `model_runner.attn_backend` by this package's
Mi355AttnBackend -- constructed from the REFERENCE model runner, called by the REFERENCE's
RadixAttention.forward with the REFERENCE's layer / ForwardBatch / MHATokenToKVPool objects -- and
runs the same requests again.  The kernels behind the backend are the tests-only CPU ops double
(there is no GPU here); what is exercised is the plugin surface: constructor, metadata hooks,
forward_extend / forward_decode, the graph hooks' signatures.

Writes tests/golden/dropin.json: the verdict of that run plus the attribute surface of every
reference object the backend was handed, so that a CPU test without the reference can rebuild
stand-ins that expose exactly those attributes and nothing else.
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import gen_golden  # noqa: E402  (installs the import harness)

import torch  # noqa: E402

import cases  # noqa: E402
import cpu_ops  # noqa: E402
from sglang_custom_amd import ops_backend  # noqa: E402
from sglang_custom_amd.layers.attention.mi355_backend import Mi355AttnBackend  # noqa: E402


def public_attrs(obj):
    return sorted(n for n in dir(obj) if not n.startswith("__"))


def main():
    from sglang.srt.layers.attention.base_attn_backend import AttentionBackend
    from sglang.srt.managers.schedule_batch import Req, ScheduleBatch
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch
    from sglang.srt.sampling.sampling_params import SamplingParams
    from sglang.srt.speculative.spec_info import SpeculativeAlgorithm

    name, preset, dtn, lens, steps = cases.E2E_CASES[0]
    mr, cfg = gen_golden.load_reference_runner(preset, dtn)
    prompts = cases.e2e_prompts(cfg["vocab_size"], lens)
    seen = {}

    def run_once():
        mr.req_to_token_pool.clear()
        mr.token_to_kv_pool_allocator.clear()
        reqs = []
        for i, ids in enumerate(prompts):
            r = Req(rid=i, origin_input_text="", origin_input_ids=list(ids),
                    sampling_params=SamplingParams(temperature=0, max_new_tokens=16))
            r.prefix_indices = []
            r.fill_ids = r.origin_input_ids
            r.extend_input_len = len(r.fill_ids)
            r.logprob_start_len = len(r.origin_input_ids) - 1
            reqs.append(r)
        batch = ScheduleBatch.init_new(
            reqs=reqs, req_to_token_pool=mr.req_to_token_pool, token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator,
            tree_cache=None, model_config=mr.model_config, enable_overlap=False,
            spec_algorithm=SpeculativeAlgorithm.NONE, enable_custom_logit_processor=False)
        toks, logits = [], []

        def step():
            fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
            seen.setdefault("forward_batch_" + fb.forward_mode.name.lower(), fb)
            lo, _ = mr.forward(fb)
            nxt = mr.sample(lo, fb)
            toks.append(nxt.clone())
            logits.append(lo.next_token_logits.clone())
            return nxt

        with torch.no_grad():
            batch.prepare_for_extend()
            nxt = step()
            for _ in range(steps - 1):
                batch.output_ids = nxt
                batch.prepare_for_decode()
                nxt = step()
        return torch.stack(toks, 1), logits

    ref_toks, ref_logits = run_once()                     # the reference's own torch-native backend
    ops_backend.set_ops_backend(cpu_ops)
    backend = Mi355AttnBackend(mr)                        # constructed from the reference's model runner
    mr.attn_backend = backend                             # the registration a maintainer adds in _get_attention_backend_from_str (model_runner.py:1313-1391)
    toks, logits = run_once()
    err = max((a - b).abs().max().item() for a, b in zip(logits, ref_logits))
    # graph hooks: the reference's CudaGraphRunner calls them positionally with these arguments
    sig = {m: list(inspect.signature(getattr(AttentionBackend, m)).parameters)
           for m in ("init_cuda_graph_state", "init_forward_metadata_capture_cuda_graph",
                     "init_forward_metadata_replay_cuda_graph", "get_cuda_graph_seq_len_fill_value",
                     "forward", "forward_decode", "forward_extend", "init_forward_metadata")}
    ours = {m: list(inspect.signature(getattr(Mi355AttnBackend, m)).parameters) for m in sig}
    layer = mr.model.model.layers[0].self_attn.attn
    out = {
        "scenario": f"{preset} {dtn}: prefill {list(lens)} + {steps - 1} greedy decode steps through the reference's "
                    "ModelRunner with model_runner.attn_backend = Mi355AttnBackend(model_runner)",
        "tokens_equal_reference_backend": bool(torch.equal(toks, ref_toks)),
        "max_abs_logit_difference": err,
        "reference_interface_signatures": sig,
        "mi355_backend_signatures": ours,
        "reference_attrs": {
            "model_runner": [a for a in ("device", "tp_size", "req_to_token_pool", "token_to_kv_pool", "model_config",
                                         "sliding_window_size", "attn_backend") if hasattr(mr, a)],
            "model_config": [a for a in ("num_attention_heads", "num_key_value_heads", "context_len", "head_dim",
                                         "v_head_dim", "vocab_size") if hasattr(mr.model_config, a)],
            "token_to_kv_pool": public_attrs(mr.token_to_kv_pool),
            "req_to_token_pool": public_attrs(mr.req_to_token_pool),
            "radix_attention": sorted(k for k in vars(layer) if not k.startswith("_")),
            "forward_batch": sorted(f for f in seen["forward_batch_extend"].__dataclass_fields__),
            # (dir() of an Enum class hides its methods: read the class dictionary)
            "forward_mode": sorted(n for n in vars(type(seen["forward_batch_extend"].forward_mode)) if not n.startswith("_")),
        },
    }
    assert out["tokens_equal_reference_backend"] and err < 1e-5, (out["tokens_equal_reference_backend"], err)
    path = os.path.join(HERE, "dropin.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, "| tokens equal:", out["tokens_equal_reference_backend"], "| max |dlogit|:", err)


if __name__ == "__main__":
    main()
