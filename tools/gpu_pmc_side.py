"""Dev aid for rocprofv3 --pmc passes over the two side configurations of BASELINE.json: the bf16 forward at
configs[1] (4,16,4096,64) and the fp8-e4m3 causal forward of configs[4] (1,16,32768,128); `n` launches each
(default 12) after a ramp at the north-star shape so that the clock is the sustained one."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mkb = lambda *s: (torch.rand(*s, device="cuda") - 0.5).bfloat16()
mk8 = lambda *s: (torch.rand(*s, device="cuda") - 0.5).to(torch.float8_e4m3fn)
Q2, K2, V2 = (mkb(4, 16, 4096, 64) for _ in range(3))
O2 = torch.empty_like(Q2); L2 = torch.empty(4, 16, 4096, device="cuda")
Q5, K5, V5 = (mk8(1, 16, 32768, 128) for _ in range(3))
O5 = torch.empty(1, 16, 32768, 128, dtype=torch.bfloat16, device="cuda"); L5 = torch.empty(1, 16, 32768, device="cuda")
Q3, K3, V3 = (mkb(4, 16, 8192, 128) for _ in range(3))
O3 = torch.empty_like(Q3); L3 = torch.empty(4, 16, 8192, device="cuda")
for _ in range(20):
    fa.flash_attention_2_forward(Q3, K3, V3, None, O=O3, L=L3)
for _ in range(n):
    fa.flash_attention_2_forward(Q2, K2, V2, None, O=O2, L=L2)
for _ in range(n):
    fa.flash_attention_2_forward(Q5, K5, V5, None, causal=True, O=O5, L=L5)
torch.cuda.synchronize()
