"""Dev aid: under `rocprofv3 --pmc GRBM_GUI_ACTIVE`, 30 back-to-back launches of the two backward forms, so that the
per-dispatch clock (GRBM_GUI_ACTIVE / 8 / duration) can be read at the sustained power state.  tools/pmc_clock_summary.py reads the CSV."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O, L = fa.flash_attention_2_forward(Q, K, V)
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
for _ in range(30):
    fa.flash_attention_2_forward(Q, K, V, O=O, L=L)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=6)
torch.cuda.synchronize()
