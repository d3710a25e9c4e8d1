import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O, L = fa.flash_attention_2_forward(Q, K, V)
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
nbytes = fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0)
ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
for _ in range(3):
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=1)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=2)
torch.cuda.synchronize()
rc = ws[nbytes // 3:].view(torch.int64)[: 2048 * 4 * 2].view(-1, 2).double().cpu()
print("per wave: compute cycles mean %.0f, sync-wait cycles mean %.0f (%.1f%%), max sync %.0f" % (
    rc[:, 0].mean(), rc[:, 1].mean(), 100 * rc[:, 1].mean() / (rc[:, 0].mean() + rc[:, 1].mean()), rc[:, 1].max()))
