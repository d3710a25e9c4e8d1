"""Dev aid for rocprofv3 --pmc passes on the fused backward prototype: two launches at the bench shape."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cuda_flashattention_amd as fa
from gpu_check_fused import fused, lib
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O, L = fa.flash_attention_2_forward(Q, K, V)
ws = torch.empty(lib.fa2_backward_fused_workspace_bytes(B, H, N, d), dtype=torch.uint8, device="cuda")
for _ in range(2): fused(Q, K, V, O, L, dO, d**-0.5, ws, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
torch.cuda.synchronize()
