import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
from cuda_flashattention_amd import ring
P, B, H, N, d = 2, 1, 2, 512, 128
g = torch.Generator().manual_seed(5)
mk = lambda: (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16().cuda()
Q, K, V = mk(), mk(), mk()
s = 1.0 / d ** 0.5
Oref, Lref = fa.flash_attention_2_forward(Q, K, V, s, causal=True)
shards = []
for r in range(P):
    rows = torch.tensor(ring.zigzag_rows(N, r, P), device="cuda")
    shards.append((rows, Q[:, :, rows].contiguous(), K[:, :, rows].contiguous(), V[:, :, rows].contiguous()))
for r in range(P):
    rows, Ql, _, _ = shards[r]
    n = Ql.shape[2]; c = n // 2
    Ol = torch.zeros_like(Ql); Ll = torch.zeros(B, H, n, device="cuda"); Ml = torch.zeros_like(Ll)
    Oacc = torch.zeros(Ql.shape, dtype=torch.float32, device="cuda")
    for step in range(P):
        owner = ring.kv_owner(r, step, P)
        kind = ring.causal_block_kind(r, owner)
        ring._gpu_block(Ql, shards[owner][2], shards[owner][3], Ol, Ll, Oacc, Ml, s, kind)
        torch.cuda.synchronize()
        print("rank", r, "step", step, "owner", owner, kind, "M[0,0,:3]", Ml[0, 0, :3].tolist(), "M[0,0,c:c+3]", Ml[0, 0, c:c + 3].tolist(),
              "l", Ll[0, 0, :2].tolist(), Ll[0, 0, c:c + 2].tolist())
    ring._gpu_finalize(Ol, Ll, Oacc, Ml)
    torch.cuda.synchronize()
    for name, sl in (("chunk0", slice(0, c)), ("chunk1", slice(c, n))):
        a = Ol[:, :, sl].float().cpu().numpy(); b = Oref[:, :, rows[sl]].float().cpu().numpy()
        la = Ll[:, :, sl].cpu().numpy(); lb = Lref[:, :, rows[sl]].cpu().numpy()
        print("  rank", r, name, "relO", np.linalg.norm(a - b) / np.linalg.norm(b), "maxdL", np.abs(la - lb).max(), "nan", int(np.isnan(a).sum()))
