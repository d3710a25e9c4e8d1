"""Ad-hoc check + timing of the causal single-kernel backward (fa2_backward, causal) against the oracle and the two-kernel form."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle
lib = fa._capi.lib()
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
f = lambda t: t.float().cpu().numpy()

def two(Q, K, V, O, L, dO, scale):
    B, H, N, d = Q.shape
    out = [torch.empty_like(Q) for _ in range(3)]
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    for ph in (1, 6):
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, causal=True, dQ=out[0], dK=out[1], dV=out[2], workspace=ws, phases=ph)
    return out

def run(B, H, N, d=128, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    scale = 1.0 / d ** 0.5
    dev = [t.cuda() for t in (Q, K, V, dO)]
    O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=True)
    got = fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=True)
    ref2 = two(dev[0], dev[1], dev[2], O, L, dev[3], scale)
    torch.cuda.synchronize()
    want = oracle.attention_backward(f(Q), f(K), f(V), f(dO), scale, causal=True)
    e = [rel(f(x), y) for x, y in zip(got, want)]
    same = [bool(torch.equal(a, b)) for a, b in zip(got, ref2)]
    print(f"B{B} H{H} N{N}: relL2 dQ={e[0]:.3e} dK={e[1]:.3e} dV={e[2]:.3e}   == two-kernel (dQ,dK,dV): {same}", flush=True)
    return max(e)

if __name__ == "__main__":
    bad = 0
    for cfg in [(1, 1, 256), (1, 2, 512), (2, 8, 1024), (1, 3, 768), (1, 9, 2048), (1, 2, 16384)]:
        bad += run(*cfg) > 8e-3
    if bad:
        sys.exit(1)
    B, H, N, d = 4, 16, 8192, 128
    mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
    Q, K, V, dO = mk(), mk(), mk(), mk()
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(Q, K, V, causal=True)
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    out = [torch.empty_like(Q) for _ in range(3)]
    a = lambda: fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, causal=True, dQ=out[0], dK=out[1], dV=out[2], workspace=ws)
    def b():
        for ph in (1, 6): fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, causal=True, dQ=out[0], dK=out[1], dV=out[2], workspace=ws, phases=ph)
    r1 = [x.clone() for x in (a() or out)]
    r2 = [x.clone() for x in (a() or out)]
    torch.cuda.synchronize()
    print("bitwise repeatable:", all(bool(torch.equal(x, y)) for x, y in zip(r1, r2)))
    def tm(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    print(f"causal (4,16,8192,128): single kernel {tm(a):.3f} ms, two kernels {tm(b):.3f} ms", flush=True)
