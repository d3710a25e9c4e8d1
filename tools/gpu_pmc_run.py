"""Dev aid for rocprofv3 --pmc passes: runs each bf16 kernel 12 times at the bench shape -- the forward, fa2_backward as
shipped (delta + the single five-product kernel + its output pass) and the two-kernel form of the backward (phases 6)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
for _ in range(12):
    fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, dQ=dQ, dK=dK, dV=dV, workspace=ws)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=6)
torch.cuda.synchronize()
