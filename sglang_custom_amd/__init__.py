"""MI355X-native (gfx950 / CDNA4) implementation of SGLang's attention-backend +
forward-batch hot path: hand-written HIP kernels behind a C ABI (libmi355_hotpath.so) and the
host-side mirror of the reference's AttentionBackend / CustomOp / ScheduleBatch ->
ForwardBatch -> ModelRunner surface."""
__version__ = "0.1.0"
