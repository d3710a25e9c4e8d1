"""Dev aid: per-kernel times at the bench shape for A/B runs of library variants (FA2_LIB_PATH): forward, fa2_backward
as shipped, its causal form, the two-kernel form; median of 5 blocks of 10 launches after a ramp."""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
fwd = lambda c=False: fa.flash_attention_2_forward(Q, K, V, None, causal=c, O=O, L=L)
bwd = lambda ph=7, c=False: fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, causal=c, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=ph)
def block(f, n=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(20): fwd(); bwd()
res = {}
for name, f in (("fwd", fwd), ("bwd", bwd), ("bwd fused only", lambda: bwd(8)), ("step", lambda: (fwd(), bwd()))):
    res[name] = statistics.median(block(f) for _ in range(5))
fwd(True)
for _ in range(5): bwd(7, True)
res["bwd causal"] = statistics.median(block(lambda: bwd(7, True)) for _ in range(5))
print("  ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
