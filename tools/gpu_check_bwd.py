"""Ad-hoc check of the bf16 backward against the oracle (dev aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def run(B, H, N, d, causal=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    scale = 1.0 / d ** 0.5
    Qd, Kd, Vd, dOd = Q.cuda(), K.cuda(), V.cuda(), dO.cuda()
    O, L = fa.flash_attention_2_forward(Qd, Kd, Vd, scale, causal=causal)
    dQ, dK, dV = fa.flash_attention_2_backward(Qd, Kd, Vd, O, L, dOd, scale, causal=causal)
    torch.cuda.synchronize()
    f = lambda t: t.float().numpy()
    rQ, rK, rV = oracle.attention_backward(f(Q), f(K), f(V), f(dO), scale, causal=causal)
    e = [rel(f(x.cpu()), y) for x, y in zip((dQ, dK, dV), (rQ, rK, rV))]
    print(f"B{B} H{H} N{N} d{d} causal={causal}: relL2 dQ={e[0]:.3e} dK={e[1]:.3e} dV={e[2]:.3e}", flush=True)
    return max(e)

if __name__ == "__main__":
    bad = 0
    for cfg in [(1,1,256,128), (1,2,128,64), (2,8,512,128), (1,8,1024,64), (1,3,333,128), (1,1,77,64),
                (2,8,512,128,True), (1,2,300,64,True), (1,4,1024,128,True)]:
        bad += run(*cfg) > 8e-3
    B,H,N,d = 4,16,8192,128
    mk = lambda: (torch.rand(B,H,N,d, device="cuda")-0.5).bfloat16()
    Q,K,V,dO = mk(),mk(),mk(),mk()
    O,L = fa.flash_attention_2_forward(Q,K,V)
    dQ,dK,dV = torch.empty_like(Q),torch.empty_like(Q),torch.empty_like(Q)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B,H,N,d,0), dtype=torch.uint8, device="cuda")
    for _ in range(2): fa.flash_attention_2_backward(Q,K,V,O,L,dO,dQ=dQ,dK=dK,dV=dV,workspace=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fa.flash_attention_2_backward(Q,K,V,O,L,dO,dQ=dQ,dK=dK,dV=dV,workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    print(f"bwd cfg3 {ms:.3f} ms  {10*B*H*N*N*d/ms/1e9:.1f} TFLOP/s (algorithmic 10N^2d)", flush=True)
    sys.exit(1 if bad else 0)
