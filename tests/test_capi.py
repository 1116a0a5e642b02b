"""The C-ABI library: builds for gfx950, loads, and exports exactly the entry points that
include/mi355_hotpath.h declares.  Argument validation happens before any HIP call, so the
error paths can be exercised without a GPU (no compute is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_hotpath.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (its HIP runtime must be the one in the process)
    from sglang_custom_amd import _lib, build

    build.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    from sglang_custom_amd import _lib

    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(_lib.EXPORTED_SYMBOLS) == decl
    for name in decl:
        assert getattr(lib, name) is not None


def test_library_exports_only_c_symbols():
    import subprocess

    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "sglang_custom_amd", "libmi355_hotpath.so")]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T mi355_" in l)
    assert exported == declared_symbols()


def test_abi_version_and_error_codes(lib):
    assert lib.mi355_abi_version() == 1
    # null pointers / bad shapes are rejected on the host with a message, nothing is launched
    assert lib.mi355_rmsnorm(None, None, None, 4, 64, 64, 64, 1e-5, 1, None) == -1
    assert b"null pointer" in lib.mi355_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mi355_rmsnorm(p, p, p, 4, 64, 32, 64, 1e-5, 1, None) == -1       # stride < hidden
    assert lib.mi355_rmsnorm(p, p, p, 4, 64, 64, 64, 1e-5, 9, None) == -2       # unknown dtype
    assert lib.mi355_decode_attention(p, p, p, p, p, 0, p, p, None, 8, None, None, None, None, 1, 2, 8, 3, 64, 64,
                                      512, 16, 0.1, 0.0, 1, None) == -1        # 8 % 3 != 0
    assert lib.mi355_decode_attention(p, p, p, p, p, 0, p, p, None, 8, None, None, None, None, 4, 2, 8, 2, 64, 64,
                                      512, 16, 0.1, 0.0, 1, None) == -1        # splits need scratch
    assert lib.mi355_linear(p, p, p, None, 4, 64, 64, 32, None, 0, 1, 1, None) == -1   # x_stride < k
    assert lib.mi355_extend_attention(p, p, p, p, p, p, p, p, p, None, None, 1, 1, 1, 8, 130, 1, 64, 64,
                                      8320, 64, 64, 16, 0.1, 0.0, -1, 1, None) == -2   # 130 heads / 1 kv
    # zero-sized work is a no-op success
    assert lib.mi355_silu_and_mul(p, p, 0, 64, 1, None) == 0
    assert lib.mi355_argmax(p, p, 0, 100, 100, None) == 0


def test_planner_is_pure_host_logic(lib):
    assert lib.mi355_decode_num_splits(64, 8, 1100, 64) == 1
    assert lib.mi355_decode_num_splits(16, 8, 1100, 64) == 1     # 128 workgroups already saturate the stream
    assert lib.mi355_decode_num_splits(64, 1, 1100, 64) == 3     # one local kv head (tensor-parallel rank of 8)
    assert lib.mi355_decode_num_splits(8, 8, 1100, 64) == 3
    assert lib.mi355_decode_num_splits(1, 8, 1100, 64) == 8
    assert lib.mi355_decode_num_splits(1, 8, 100000, 16) == 16
    assert lib.mi355_decode_num_splits(1, 1, 4104, 64) == 19     # one sequence, one kv head: sqrt(keys / 12), not keys / 128
    assert lib.mi355_decode_num_splits(256, 8, 50, 64) == 1
    # o_proj at batch 64: 64 column blocks x 4 K-splits -> 4 KiB of arrival counters + fp32 partials
    assert lib.mi355_linear_workspace_bytes(64, 4096, 4096) == 4096 + 4 * 64 * 4096 * 4
    # partial tiles are whole 16-row MFMA tiles
    assert lib.mi355_linear_workspace_bytes(5, 4096, 4096) == 4096 + 4 * 16 * 4096 * 4
    assert lib.mi355_linear_workspace_bytes(64, 28672, 4096) == 0    # wide outputs are never split
    assert lib.mi355_linear_workspace_bytes(64, 128256, 4096) == 0
    assert lib.mi355_linear_workspace_bytes(65, 4096, 4096) == 0
