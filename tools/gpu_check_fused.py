"""Ad-hoc check + timing of the fused five-product backward prototype (fa2_backward_fused) against the oracle and the
two-kernel backward (dev aid)."""
import sys
import ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle

lib = fa._capi.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1

def fused(Q, K, V, O, L, dO, scale, ws=None, mode=None):
    B, H, N, d = Q.shape
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    nb = lib.fa2_backward_fused_workspace_bytes(B, H, N, d)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda") if ws is None else ws
    st = lib.fa2_backward_fused(P(Q), P(K), P(V), P(O), P(L), P(dO), P(dQ), P(dK), P(dV), B, H, N, d, scale, MODE if mode is None else mode, P(ws), nb,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
    return dQ, dK, dV

def run(B, H, N, d=128, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    scale = 1.0 / d ** 0.5
    Qd, Kd, Vd, dOd = Q.cuda(), K.cuda(), V.cuda(), dO.cuda()
    O, L = fa.flash_attention_2_forward(Qd, Kd, Vd, scale)
    dQ, dK, dV = fused(Qd, Kd, Vd, O, L, dOd, scale)
    torch.cuda.synchronize()
    f = lambda t: t.float().numpy()
    rQ, rK, rV = oracle.attention_backward(f(Q), f(K), f(V), f(dO), scale)
    e = [rel(f(x.cpu()), y) for x, y in zip((dQ, dK, dV), (rQ, rK, rV))]
    print(f"B{B} H{H} N{N} d{d}: relL2 dQ={e[0]:.3e} dK={e[1]:.3e} dV={e[2]:.3e}", flush=True)
    return max(e)

if __name__ == "__main__":
    bad = 0
    for cfg in [(1, 1, 256), (1, 2, 512), (2, 8, 1024), (1, 3, 768), (1, 9, 2048), (1, 70, 256), (1, 2, 16384)]:
        bad += run(*cfg) > 8e-3
    if bad:
        sys.exit(1)
    B, H, N, d = 4, 16, 8192, 128
    mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
    Q, K, V, dO = mk(), mk(), mk(), mk()
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(Q, K, V)
    ref = [torch.empty_like(Q) for _ in range(3)]
    ws0 = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    for ph in (1, 6): fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, dQ=ref[0], dK=ref[1], dV=ref[2], workspace=ws0, phases=ph)
    nb = lib.fa2_backward_fused_workspace_bytes(B, H, N, d)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    got = fused(Q, K, V, O, L, dO, scale, ws)
    torch.cuda.synchronize()
    again = fused(Q, K, V, O, L, dO, scale, ws)
    torch.cuda.synchronize()
    print("mode", MODE, "bitwise repeatable:", all(bool((a == b).all()) for a, b in zip(got, again)), flush=True)
    for n, a, b in zip("QKV", got, ref):
        print(f"cfg3 d{n}: relL2 vs two-kernel {float((a.float()-b.float()).norm()/b.float().norm()):.3e}", flush=True)
    for _ in range(3): fused(Q, K, V, O, L, dO, scale, ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fused(Q, K, V, O, L, dO, scale, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fused bwd cfg3 {ms:.3f} ms  {10*B*H*N*N*d/ms/1e9:.1f} TFLOP/s (algorithmic 10N^2d)", flush=True)
