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
    text = re.sub(r"^[ \t]*#[^\n]*(\\\n[^\n]*)*", "", text, flags=re.M)      # preprocessor lines (function-like macros are not exports)
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
    # d = 64 (round 4): aligned lengths run the single kernel too -- the same running-sum layout (128 floats per row: four waves x
    # 32 columns, two key halves per column block) and control block as d = 128; other lengths the two kernels
    assert lib.fa2_backward_workspace_bytes(4, 16, 8192, 64, 0) == lib.fa2_backward_workspace_bytes(4, 16, 8192, 128, 0)
    assert lib.fa2_backward_workspace_bytes(4, 16, 800, 64, 0) == 3 * ((4 * 16 * 800 * 4 + 255) // 256 * 256)      # 1024 against 832: two kernels
    assert lib.fa2_backward_workspace_bytes(4, 16, 8192, 128, 1) == base         # fp32
    fused = lib.fa2_backward_workspace_bytes(4, 16, 8192, 128, 0)                # + fp32 dQ sums + a control block
    assert base + 4 * 16 * 8192 * 128 * 4 < fused < base + 4 * 16 * 8192 * 128 * 4 + (1 << 20)
    assert lib.fa2_backward_fused_workspace_bytes(4, 16, 8192, 128) == fused
    # ragged seq_len: the single kernel pads to a multiple of 256 and is taken when 5 roundup(N, 256) <= 7 roundup(N, 64)
    assert lib.fa2_backward_workspace_bytes(4, 16, 300, 128, 0) == 3 * ((4 * 16 * 300 * 4 + 255) // 256 * 256)     # 2560 > 2240: two kernels
    ragged = lib.fa2_backward_workspace_bytes(4, 16, 8000, 128, 0)                                                 # 40960 <= 56000: single
    assert ragged > 3 * ((4 * 16 * 8000 * 4 + 255) // 256 * 256) + 4 * 16 * 8192 * 128 * 4 + 2 * 4 * 16 * 8192 * 4
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


def test_product_library_has_no_test_hooks():
    """Fault injection and the forced grid size exist only in tests/loopback/libfa2_mi355x_hooks.so (fa2_bwd_fused.hip built
    -DFA2_TEST_HOOKS); the product library neither exports the setter nor reads the round-2 environment switches, and the
    only environment variable it reads at all is the implementation selector FA2_BACKWARD_PATH (once, cached; the round-3
    FA2_FORWARD_PATH went with the round-2 forward kernel it selected)."""
    so = os.path.join(ROOT, "cuda_flashattention_amd", "lib", "libfa2_mi355x.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert "fa2_test_set_fused_hooks" not in exported
    strings = subprocess.run(["strings", so], capture_output=True, text=True, check=True).stdout
    assert "FA2_FUSED_FAULT" not in strings and "FA2_FUSED_GRID" not in strings
    src = os.path.join(ROOT, "cuda_flashattention_amd", "csrc")
    hits = subprocess.run(["grep", "-rn", "getenv", src, "--include=*.hip", "--include=*.cpp", "--include=*.h"],
                          capture_output=True, text=True).stdout.strip().splitlines()
    assert len(hits) == 1 and "FA2_BACKWARD_PATH" in hits[0], hits
    assert "FA2_FORWARD_PATH" not in strings
    hooks = os.path.join(ROOT, "tests", "loopback", "libfa2_mi355x_hooks.so")
    if os.path.exists(hooks):
        out = subprocess.run(["nm", "-D", "--defined-only", hooks], capture_output=True, text=True, check=True).stdout
        assert "fa2_test_set_fused_hooks" in out


def test_plan_and_status_argument_checks():
    """fa2_backward_plan / fa2_backward_status validate before touching a device (no GPU here: only the early returns)."""
    from cuda_flashattention_amd import _capi
    lib = _capi.lib()
    why = ctypes.c_char_p()
    assert lib.fa2_backward_plan(0, 1, 256, 128, 0, 0, ctypes.byref(why)) == -2
    assert lib.fa2_backward_plan(1, 1, 256, 128, 2, 0, ctypes.byref(why)) == -4          # fp8 has no backward
    assert lib.fa2_backward_plan(1, 1, 256, 96, 0, 0, ctypes.byref(why)) == -3
    assert lib.fa2_backward_plan(1, 1, 300, 128, 0, 0, ctypes.byref(why)) == 2 and b"multiple of 256" in why.value
    assert lib.fa2_backward_plan(1, 1, 256, 64, 1, 0, ctypes.byref(why)) == 2 and b"fp32" in why.value
    assert lib.fa2_backward_status(None, 0, 1, 1, 256, 128, 0, None) == -1
    assert b"NaN" in lib.fa2_status_string(-7)
    # forward-only problems are no longer limited by the backward's row-constant planes (B H N 8 bytes < 2 GiB)
    one = ctypes.c_void_p(16)
    assert lib.fa2_forward(one, one, one, one, one, 64, 64, 1 << 17, 64, 0.125, 7, 0, None) == -4    # passes the shape checks
    assert lib.fa2_backward(*([one] * 9), 64, 64, 1 << 17, 64, 0.125, 0, 0, one, 1 << 40, None) == -2
