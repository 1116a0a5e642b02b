"""TEST INFRASTRUCTURE -- CPU restatement of the reference's hot path (the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only to CHECK or to time the CPU baseline: nothing under sglang_custom_amd/ imports it, and
the product path has no CPU fallback.

Every function cites the reference file:line it restates (paths relative to the reference
checkout).  Parity pinning: the functions here are checked against golden vectors produced by
the reference's own Python path and its sgl-kernel CPU kernels (tests/golden/*.npz, generated
by tests/golden/gen_golden.py in the build container where the reference is importable).
"""
