"""Dev aid (experiment): what the single-kernel backward gains when the multiply in front of each exponential goes away.
The variant library's bodies compute P = exp2(S') with no scaling; fed K' = c2 K and L' = c2 L (c2 = scale log2 e) it
forms the same P as the product on (K, L) -- same data, same power -- so the two timings compare like for like
(its dQ / dK come out differently scaled: timing only).  usage: [FA2_LIB_PATH=var/ks_exp.so] python tools/gpu_ab_kscaled.py [scaled]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
g = torch.Generator(device="cuda").manual_seed(1)
mk = lambda s: ((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * s).bfloat16()
Q, K, V, dO = mk(1.0), mk(1.0), mk(1.0), mk(0.4)
scale = d ** -0.5
O, L = fa.flash_attention_2_forward(Q, K, V, scale)
if len(sys.argv) > 1 and sys.argv[1] == "scaled":
    c2 = scale * 1.4426950408889634
    K = (K.float() * c2).bfloat16()
    L = L * c2
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
f = lambda: fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, dQ=dQ, dK=dK, dV=dV, workspace=ws)
for _ in range(30): f()
torch.cuda.synchronize()
v = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    v.append(e0.elapsed_time(e1) / 20)
print(f"{os.environ.get('FA2_LIB_PATH', 'product'):18s} {'scaled inputs' if len(sys.argv) > 1 else 'plain inputs ':13s} backward {sorted(v)[2]:.4f} ms (min {min(v):.4f})  finite dK: {bool(torch.isfinite(dK.float()).all())}", flush=True)
