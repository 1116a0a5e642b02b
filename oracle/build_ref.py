"""Recipe: compile the reference's own C++ CPU kernels (sgl-kernel/csrc/cpu/*.cpp) where they
lie under /root/reference into oracle/_ref/common_ops.so.

TEST INFRASTRUCTURE ONLY.  The resulting library registers torch.ops.sgl_kernel.*_cpu ops
(decode_attention_cpu, extend_attention_cpu, rmsnorm_cpu, fused_add_rmsnorm_cpu,
silu_and_mul_cpu, rotary_embedding_cpu, ...: sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:232-363).
It is used (a) to validate the numpy/torch restatement in oracle/, (b) by the golden-vector
generator, which needs these ops present before the reference's Python imports.

No reference source is copied: g++ reads the files in place; only the .so (+ a sidecar json with
the CPU flags it was built for) is written, under oracle/_ref/, which is git-ignored.
The reference's CMake build is NOT run; the flags below restate its add_compile_options
(sgl-kernel/csrc/cpu/CMakeLists.txt:66-76).
"""
import glob
import json
import os
import shutil
import sys

REF = os.environ.get("SGLANG_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def cpu_flags():
    with open("/proc/cpuinfo") as f:
        for line in f:
            if line.startswith("flags"):
                return sorted(line.split(":", 1)[1].split())
    return []


def build(verbose=False):
    src_dir = os.path.join(REF, "sgl-kernel", "csrc", "cpu")
    if not os.path.isdir(src_dir):
        raise FileNotFoundError(f"reference sources not found at {src_dir}")
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "common_ops.so")
    if os.path.exists(so):
        return so
    from torch.utils.cpp_extension import load

    build_dir = os.path.join(OUT, "build")
    os.makedirs(build_dir, exist_ok=True)
    load(
        name="common_ops",
        sources=sorted(glob.glob(os.path.join(src_dir, "*.cpp"))),
        extra_include_paths=[
            os.path.join(REF, "sgl-kernel", "include"),
            os.path.join(REF, "sgl-kernel", "csrc"),
        ],
        extra_cflags=["-O3", "-march=native", "-fopenmp", "-Wno-unknown-pragmas",
                      "-DSGLANG_CPU_FP8_CVT_FTZ", "-std=c++17"],
        extra_ldflags=["-lnuma", "-fopenmp"],
        build_directory=build_dir,
        is_python_module=False,
        verbose=verbose,
    )
    shutil.copy(os.path.join(build_dir, "common_ops.so"), so)
    with open(os.path.join(OUT, "common_ops.json"), "w") as f:
        json.dump({"built_for_cpu_flags": cpu_flags()}, f)
    shutil.rmtree(build_dir, ignore_errors=True)
    return so


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
