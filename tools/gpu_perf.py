"""Dev aid: per-kernel timings at the bench shape + a quick parity spot check."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle

def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
f = lambda t: t.float().cpu().numpy()

def spot(B, H, N, d, causal):
    g = torch.Generator().manual_seed(3)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    s = 1.0 / d ** 0.5
    Qd, Kd, Vd, Gd = Q.cuda(), K.cuda(), V.cuda(), dO.cuda()
    O, L = fa.flash_attention_2_forward(Qd, Kd, Vd, s, causal=causal)
    dQ, dK, dV = fa.flash_attention_2_backward(Qd, Kd, Vd, O, L, Gd, s, causal=causal)
    torch.cuda.synchronize()
    Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
    gr = oracle.attention_backward(f(Q), f(K), f(V), f(dO), s, causal=causal)
    e = [rel(f(O), Or)] + [rel(f(x), y) for x, y in zip((dQ, dK, dV), gr)]
    ok = max(e) < 5e-3
    print(f"spot B{B} H{H} N{N} d{d} c={int(causal)}: O {e[0]:.2e} dQ {e[1]:.2e} dK {e[2]:.2e} dV {e[3]:.2e} {'ok' if ok else 'BAD'}", flush=True)
    return ok

def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

if __name__ == "__main__":
    ok = all([spot(1, 8, 640, 128, False), spot(1, 3, 333, 128, True), spot(1, 8, 320, 64, False), spot(1, 2, 300, 64, True)])
    shapes = [(4, 16, 8192, 128, False)]
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        shapes += [(4, 16, 4096, 64, False), (1, 16, 8192, 128, True)]
    for (B, H, N, d, causal) in shapes:
        mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
        Q, K, V, dO = mk(), mk(), mk(), mk()
        O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
        ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
        fwd = lambda: fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L)
        bwd = lambda ph: fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, causal=causal, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=ph)
        fwd(); bwd(7)
        tf, t1, t2, t4 = timeit(fwd), timeit(lambda: bwd(1)), timeit(lambda: bwd(2)), timeit(lambda: bwd(4))
        unit = 2.0 * B * H * N * N * d * (0.5 if causal else 1.0) / 1e9
        tot = tf + t1 + t2 + t4
        print(f"B{B} H{H} N{N} d{d} c={int(causal)}: fwd {tf:.3f} ms ({2*unit/tf:.0f} TF) | delta {t1:.3f} | dq {t2:.3f} ms ({3*unit/t2:.0f} TF exec) | "
              f"dkdv {t4:.3f} ms ({4*unit/t4:.0f} TF exec) | fwd+bwd {tot:.3f} ms = {7*unit/tot:.0f} TF algorithmic", flush=True)
    sys.exit(0 if ok else 1)
