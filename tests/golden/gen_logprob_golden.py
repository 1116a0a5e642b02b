"""Generate tests/golden/logprob.json: input-token log-probabilities of a prefill, from the REFERENCE's own
ScheduleBatch.prepare_for_extend (logprob bookkeeping, managers/schedule_batch.py:1148-1272) and
LogitsProcessor.forward (the extend_return_logprob branch, layers/logits_processor.py:274-420) through its
ModelRunner on CPU.  Build container only:  python tests/golden/gen_logprob_golden.py
Two scenarios on the tiny fp32 Llama: a plain prefill with per-request logprob_start_len / top_logprobs_num,
and an extend over a cached prefix (logprob_start_len before, inside and after the prefix)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the import harness)

import torch  # noqa: E402

import cases  # noqa: E402


def main():
    from sglang.srt.managers.schedule_batch import Req, ScheduleBatch
    from sglang.srt.model_executor.forward_batch_info import ForwardBatch
    from sglang.srt.sampling.sampling_params import SamplingParams
    from sglang.srt.speculative.spec_info import SpeculativeAlgorithm

    preset, dtn = "tiny-llama-d64", "f32"
    mr, cfg = G.load_reference_runner(preset, dtn)

    def new_batch(reqs):
        return ScheduleBatch.init_new(
            reqs=reqs, req_to_token_pool=mr.req_to_token_pool,
            token_to_kv_pool_allocator=mr.token_to_kv_pool_allocator, tree_cache=None,
            model_config=mr.model_config, enable_overlap=False,
            spec_algorithm=SpeculativeAlgorithm.NONE, enable_custom_logit_processor=False)

    def mk_req(i, ids, start, topk):
        r = Req(rid=i, origin_input_text="", origin_input_ids=list(ids),
                sampling_params=SamplingParams(temperature=0, max_new_tokens=16),
                return_logprob=True, top_logprobs_num=topk)
        r.prefix_indices = []
        r.fill_ids = r.origin_input_ids
        r.extend_input_len = len(r.fill_ids)
        r.logprob_start_len = start
        return r

    def run(batch):
        fb = ForwardBatch.init_new(batch.get_model_worker_batch(), mr)
        lo, _ = mr.forward(fb)
        return lo, fb

    def record(lo, batch):
        return dict(
            extend_logprob_start_lens=list(batch.extend_logprob_start_lens),
            extend_input_logprob_token_ids=batch.extend_input_logprob_token_ids.tolist(),
            next_token_logits_top1=lo.next_token_logits.argmax(-1).tolist(),
            next_token_logits_max=lo.next_token_logits.max(-1).values.tolist(),
            input_token_logprobs=lo.input_token_logprobs.tolist(),
            input_top_logprobs_val=lo.input_top_logprobs_val, input_top_logprobs_idx=lo.input_top_logprobs_idx)

    lens = (5, 17, 32)
    prompts = cases.e2e_prompts(cfg["vocab_size"], lens)
    out = {"preset": preset, "dtype": dtn, "lens": list(lens)}
    with torch.no_grad():
        # scenario 1: plain prefill
        starts, topks = [0, 3, lens[2] - 1], [2, 0, 3]
        mr.req_to_token_pool.clear()
        mr.token_to_kv_pool_allocator.clear()
        batch = new_batch([mk_req(i, p, starts[i], topks[i]) for i, p in enumerate(prompts)])
        batch.prepare_for_extend()
        lo, _ = run(batch)
        out["prefill"] = dict(logprob_start_lens=starts, top_logprobs_nums=topks, **record(lo, batch))
        # scenario 2: extend over a cached prefix of `cut` tokens
        cut = 4
        starts2, topks2 = [0, cut + 2, 40], [1, 2, 0]   # before the prefix, inside the new part, past the end
        mr.req_to_token_pool.clear()
        mr.token_to_kv_pool_allocator.clear()
        reqs = [mk_req(i, p[:cut], cut - 1, 0) for i, p in enumerate(prompts)]
        for r in reqs:
            r.return_logprob = False
        batch = new_batch(reqs)
        batch.prepare_for_extend()
        run(batch)
        for i, r in enumerate(reqs):
            r.fill_ids = r.fill_ids + prompts[i][cut:]
            r.origin_input_ids = list(r.fill_ids)
            r.prefix_indices = mr.req_to_token_pool.req_to_token[i, :cut]
            r.extend_input_len = len(r.fill_ids) - len(r.prefix_indices)
            r.return_logprob, r.logprob_start_len, r.top_logprobs_num = True, starts2[i], topks2[i]
        mr.req_to_token_pool.clear()
        batch = new_batch(reqs)
        batch.prepare_for_extend()
        lo, _ = run(batch)
        out["prefix_extend"] = dict(cut=cut, logprob_start_lens=starts2, top_logprobs_nums=topks2, **record(lo, batch))
    path = os.path.join(HERE, "logprob.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
