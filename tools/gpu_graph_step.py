"""Dev probe: does replaying fa2_forward + fa2_backward as ONE captured HIP graph shorten the step (the gaps between its four
kernels and one memset), against eager launches through the C ABI?  (4,16,8192,128), same process, alternating blocks."""
import sys, statistics, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda s=1.0: ((torch.rand(B, H, N, d, device="cuda") - 0.5) * s).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk(0.4)
O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
def step():
    fa.flash_attention_2_forward(Q, K, V, O=O, L=L)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws)
for _ in range(5): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step()
torch.cuda.synchronize()
def block(f, n=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a, b = [], []
for _ in range(7):
    a.append(block(step)); b.append(block(g.replay))
print(f"eager {statistics.median(a):.4f} ms/step   graph replay {statistics.median(b):.4f} ms/step   ({statistics.median(b)/statistics.median(a):.4f})")
import time
for t in (O, dQ, dK, dV): t.zero_()
step(); torch.cuda.synchronize()
ref = [t.clone() for t in (O, dQ, dK, dV)]
for t in (O, dQ, dK, dV): t.zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay reproduces the eager results:", all(torch.equal(a, b) for a, b in zip(ref, (O, dQ, dK, dV))))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"wall clock: graph {(t1 - t0) / 50 * 1e3:.4f} ms/step, eager {(t2 - t1) / 50 * 1e3:.4f} ms/step")
