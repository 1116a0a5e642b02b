"""Build libmi355_hotpath.so (the C-ABI library of HIP kernels) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the source tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355_hotpath.so")
# the development build: the same sources with -DMI355_DEV_BUILD (ablation switches become run-time
# variables set through mi355_debug_set) plus the bandwidth probes; used by tools/, never by the package
LIB_DEV = os.path.join(HERE, "libmi355_hotpath_dev.so")
SOURCES = ["elementwise.hip", "index_ops.hip", "attn_decode.hip", "attn_extend.hip", "gemm.hip",
           "gemm_tiled.hip", "sampling.hip", "quant_fp8.hip", "allreduce.hip", "mla_proj.hip"]
DEV_SOURCES = ["debug_bw.hip"]
HEADERS = [os.path.join(CSRC, "common.cuh"),
           os.path.join(os.path.dirname(HERE), "include", "mi355_hotpath.h"),
           os.path.join(os.path.dirname(HERE), "include", "mi355_hotpath_dev.h")]
# -amdgpu-mfma-vgpr-form: matrix-core accumulators stay in ordinary vector registers.  The kernels
# here run softmax / rescale / epilogue arithmetic on them between MFMAs; with the default AGPR
# form every such use costs a v_accvgpr_read/_write pair (hundreds per loop body in the attention
# kernels), and no kernel needs more than the 256 registers the VGPR form can address.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build_dev" if dev else "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES + (DEV_SOURCES if dev else []) if os.path.exists(os.path.join(CSRC, s))]
    lib = LIB_DEV if dev else LIB
    flags = FLAGS + (["-DMI355_DEV_BUILD"] + os.environ.get("MI355_DEV_EXTRA_FLAGS", "").split() if dev else [])
    objs, procs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {s} ---\n{out.decode()}\n")
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
