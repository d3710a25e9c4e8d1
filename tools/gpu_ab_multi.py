"""A/B/C... timing of several builds of libfa2_mi355x.so in ONE process, interleaved rounds (dev aid; rule 24 of the
programming guide: perf deltas come from interleaved rounds in one process on one device).

    python tools/gpu_ab_multi.py OP [--rounds R] [--iters I] [--causal] name=path.so [name=path.so ...]

OP: bwd   fa2_backward_phases(phases = 8) at (4,16,8192,128): the single-kernel backward + its output pass
    fwd   fa2_forward bf16 at (4,16,8192,128)
    fwd64 fa2_forward bf16 at (4,16,4096,64)
    fp8   fa2_forward_fp8 (caller workspace) at (1,16,32768,128), causal
    step  fa2_forward + fa2_backward at (4,16,8192,128)
The libraries must be linked -Bsymbolic (tools/build_*_variant.sh do) so that each binds its own kernels.  The first
library is the reference for the output comparison (max |diff| of every output; ablation builds are expected to differ).
Prints per library: median / min / max ms over the rounds, and the ratio of medians to the first one."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_flashattention_amd import _capi  # noqa: E402  (signatures only; the product library is NOT loaded by this import)


def load(path):
    h = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in _capi.SIGNATURES.items():
        if not hasattr(h, name):          # an older variant build without an entry point added since
            continue
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("op")
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--shape", type=str, default="")
    a = ap.parse_args()
    libs = [(s.split("=", 1)[0], load(s.split("=", 1)[1])) for s in a.libs]
    dev = torch.device("cuda")
    op = a.op
    B, H, N, d = {"bwd": (4, 16, 8192, 128), "fwd": (4, 16, 8192, 128), "fwd64": (4, 16, 4096, 64), "fp8": (1, 16, 32768, 128),
                  "step": (4, 16, 8192, 128)}[op]
    if a.shape:
        B, H, N, d = (int(x) for x in a.shape.split(","))
    causal = 1 if (a.causal or op == "fp8") else 0
    g = torch.Generator(device=dev).manual_seed(1234)
    dt = torch.float8_e4m3fn if op == "fp8" else torch.bfloat16
    mk = lambda s: ((torch.rand(B, H, N, d, device=dev, generator=g) - 0.5) * s).to(dt)
    Q, K, V = mk(1.0), mk(1.0), mk(1.0)
    dO = ((torch.rand(B, H, N, d, device=dev, generator=g) - 0.5) * 0.4).to(torch.bfloat16)
    scale = 1.0 / d ** 0.5
    cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    code = 2 if op == "fp8" else 0
    outs = {}
    calls = {}
    l0 = libs[0][1]
    O0 = torch.empty(B, H, N, d, dtype=torch.bfloat16, device=dev)
    L0 = torch.empty(B, H, N, dtype=torch.float32, device=dev)
    if op in ("bwd", "step"):
        assert l0.fa2_forward(P(Q), P(K), P(V), P(O0), P(L0), B, H, N, d, scale, 0, causal, cs) == 0
    for name, l in libs:
        if op in ("bwd", "step"):
            nb = l.fa2_backward_workspace_bytes(B, H, N, d, 0)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            dQ, dK, dV = (torch.empty_like(Q) for _ in range(3))
            O, L = (O0, L0) if op == "bwd" else (torch.empty_like(O0), torch.empty_like(L0))
            def bwd(ph, l=l, ws=ws, nb=nb, dQ=dQ, dK=dK, dV=dV, O=O, L=L):
                st = l.fa2_backward_phases(P(Q), P(K), P(V), P(O), P(L), P(dO), P(dQ), P(dK), P(dV), B, H, N, d, scale, 0, causal, P(ws), nb, cs, ph)
                assert st == 0, st
            bwd(1)
            if op == "bwd":
                calls[name] = lambda bwd=bwd: bwd(8)
                outs[name] = (dQ, dK, dV)
            else:
                def step(l=l, bwd=bwd, O=O, L=L):
                    assert l.fa2_forward(P(Q), P(K), P(V), P(O), P(L), B, H, N, d, scale, 0, causal, cs) == 0
                    bwd(7)
                calls[name] = step
                outs[name] = (O, L, dQ, dK, dV)
        elif op in ("fwd", "fwd64"):
            O, L = torch.empty_like(O0), torch.empty_like(L0)
            def fwd(l=l, O=O, L=L):
                assert l.fa2_forward(P(Q), P(K), P(V), P(O), P(L), B, H, N, d, scale, code, causal, cs) == 0
            calls[name] = fwd
            outs[name] = (O, L)
        elif op == "fp8":
            O, L = torch.empty_like(O0), torch.empty_like(L0)
            nb = l.fa2_forward_fp8_workspace_bytes(B, H, N, d)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            def f8(l=l, O=O, L=L, ws=ws, nb=nb):
                assert l.fa2_forward_fp8(P(Q), P(K), P(V), P(O), P(L), B, H, N, d, scale, causal, P(ws), nb, cs) == 0
            calls[name] = f8
            outs[name] = (O, L)
    names = [n for n, _ in libs]
    for n in names:                    # code-object load, clock ramp
        for _ in range(5):
            calls[n]()
    torch.cuda.synchronize()
    ref = [t.float().clone() for t in outs[names[0]]]
    for n in names[1:]:
        diffs = [float((t.float() - r).abs().max()) for t, r in zip(outs[n], ref)]
        bad = [not bool(torch.isfinite(t.float()).all()) for t in outs[n]]
        print(f"{n}: max|diff| vs {names[0]}: " + " ".join(f"{x:.3e}" for x in diffs) + ("  NON-FINITE" if any(bad) else ""), flush=True)
    times = {n: [] for n in names}
    clocks = {n: [] for n in names}
    # mean shader clock over each timed block (fa2_read_clocks of the LAST library, which is built from the current source:
    # per XCC d(s_memtime) / d(s_memrealtime) x 100 MHz, mean over the XCCs) -- separates "fewer cycles" from "a higher clock"
    rc = libs[-1][1].fa2_read_clocks if hasattr(libs[-1][1], "fa2_read_clocks") else None
    for r in range(a.rounds):
        order = names if r % 2 == 0 else names[::-1]
        for n in order:
            c0 = torch.zeros(16, 2, dtype=torch.int64, device=dev)
            c1 = torch.zeros(16, 2, dtype=torch.int64, device=dev)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            if rc:
                rc(P(c0), cs)
            e0.record()
            for _ in range(a.iters):
                calls[n]()
            e1.record()
            if rc:
                rc(P(c1), cs)
            e1.synchronize()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / a.iters)
            if rc:
                x, y = c0.cpu().tolist(), c1.cpu().tolist()
                v = [(q[0] - p_[0]) / (q[1] - p_[1]) * 100.0 for p_, q in zip(x, y) if p_[1] and q[1] > p_[1]]
                if v:
                    clocks[n].append(sum(v) / len(v))
    base = statistics.median(times[names[0]])
    print(f"op {op} shape ({B},{H},{N},{d}) causal {causal}: {a.rounds} rounds x {a.iters} launches, interleaved")
    for n in names:
        t = times[n]
        med = statistics.median(t)
        ck = f"  clock {statistics.median(clocks[n]):7.1f} MHz  Mcycles {med * statistics.median(clocks[n]) / 1e3:7.3f}" if clocks[n] else ""
        print(f"  {n:14s} median {med:.4f} ms  min {min(t):.4f}  max {max(t):.4f}   x{med / base:.4f} of {names[0]}{ck}", flush=True)


if __name__ == "__main__":
    main()
