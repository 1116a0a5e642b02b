"""SamplingParams normalisation rules of python/sglang/srt/sampling/sampling_params.py:18-19,
85-90: temperature < 1e-6 means greedy and is rewritten to (temperature=1, top_k=1);
top_k = -1 means the whole vocabulary (1 << 30)."""
_SAMPLING_EPS = 1e-6
TOP_K_ALL = 1 << 30


class SamplingParams:
    def __init__(self, max_new_tokens: int = 128, temperature: float = 1.0, top_p: float = 1.0,
                 top_k: int = -1, min_p: float = 0.0, ignore_eos: bool = True):
        self.max_new_tokens = max_new_tokens
        self.temperature = temperature
        self.top_p = top_p
        self.top_k = top_k
        self.min_p = min_p
        self.ignore_eos = ignore_eos
        if 0 <= self.temperature < _SAMPLING_EPS:
            self.temperature = 1.0
            self.top_k = 1
        if self.top_k == -1:
            self.top_k = TOP_K_ALL
