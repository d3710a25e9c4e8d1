"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute
calls without a GPU), argument checking returns the documented status codes, and the
product package never routes through the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)    # callback tables are not exports
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)))


def test_header_symbols_exported():
    names = _declared("fa2_mi355x.h")
    assert "flash_attention_2_forward" in names and "flash_attention_2_backward" in names
    assert "fa2_forward" in names and "fa2_backward" in names and "fa2_forward_step" in names
    lib = ctypes.CDLL(os.path.join(ROOT, "cuda_flashattention_amd", "lib", "libfa2_mi355x.so"))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fa2_mi355x.h but not exported"


def test_binding_table_matches_header():
    from cuda_flashattention_amd import _capi
    assert sorted(_capi.SIGNATURES) == _declared("fa2_mi355x.h")
    _capi.lib()


def test_ring_header_symbols_exported():
    names = _declared("fa2_ring_mi355x.h")
    assert "ring_attention_forward" in names and "fa2_ring_attention_forward" in names
    out = subprocess.run(["nm", "-D", "--defined-only",
                          os.path.join(ROOT, "cuda_flashattention_amd", "lib", "libfa2_ring_mi355x.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for n in names:
        assert n in exported, f"{n} declared in include/fa2_ring_mi355x.h but not exported"
    from cuda_flashattention_amd import ring
    assert sorted(ring.RING_SIGNATURES) == names


def test_argument_checking_status_codes():
    """Status codes instead of the reference's assert/exit (flash_attention_kernel.cu:317)."""
    from cuda_flashattention_amd import _capi
    lib = _capi.lib()
    assert lib.fa2_forward(None, None, None, None, None, 1, 1, 128, 64, 0.125, 0, 0, None) == -1
    one = ctypes.c_void_p(16)   # never dereferenced: validation fails first
    assert lib.fa2_forward(one, one, one, one, one, 1, 1, 128, 96, 0.125, 0, 0, None) == -3   # bf16 d=96
    assert lib.fa2_forward(one, one, one, one, one, 1, 1, 128, 200, 0.125, 1, 0, None) == -3  # f32 d>128
    assert lib.fa2_forward(one, one, one, one, one, 1, 1, 0, 64, 0.125, 0, 0, None) == -2
    assert lib.fa2_forward(one, one, one, one, one, 1, 1, 128, 64, -1.0, 0, 0, None) == -2
    assert lib.fa2_forward(one, one, one, one, one, 1, 1, 128, 64, 0.125, 7, 0, None) == -4
    assert lib.fa2_backward(*([one] * 9), 1, 1, 128, 64, 0.125, 0, 0, None, 0, None) == -5
    base = 3 * 4 * 16 * 8192 * 4                       # D and the two row-constant planes
    assert lib.fa2_backward_workspace_bytes(4, 16, 8192, 64, 0) == base          # d = 64: the two-kernel form only
    assert lib.fa2_backward_workspace_bytes(4, 16, 8192, 128, 1) == base         # fp32
    fused = lib.fa2_backward_workspace_bytes(4, 16, 8192, 128, 0)                # + fp32 dQ sums + a control block
    assert base + 4 * 16 * 8192 * 128 * 4 < fused < base + 4 * 16 * 8192 * 128 * 4 + (1 << 20)
    assert lib.fa2_backward_fused_workspace_bytes(4, 16, 8192, 128) == fused
    assert lib.fa2_backward_workspace_bytes(4, 16, 8000, 128, 0) == 3 * ((4 * 16 * 8000 * 4 + 255) // 256 * 256)   # N % 256 != 0
    assert b"head_dim" in lib.fa2_status_string(-3)


def test_product_never_touches_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "cuda_flashattention_amd")
    hits = subprocess.run(["grep", "-rIl", "-E", r"oracle|naive_attention\.c", pkg,
                           "--include=*.py", "--include=*.cpp", "--include=*.hip", "--include=*.h",
                           "--include=Makefile"], capture_output=True, text=True).stdout.split()
    assert hits == [], hits
    out = subprocess.run(["ldd", os.path.join(pkg, "lib", "libfa2_mi355x.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
