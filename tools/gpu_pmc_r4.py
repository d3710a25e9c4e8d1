"""Dev aid for rocprofv3 --pmc passes (round 4): 12 launches each of the kernels that changed or appeared this round and are
not in tools/gpu_pmc_run.py -- the causal forward / backward at the bench shape and the head_dim-64 backward (single kernel
and, beside it, the two kernels)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
lib = fa._capi.lib()
def run(B, H, N, d, causal, two=False):
    mk = lambda s=1.0: ((torch.rand(B, H, N, d, device="cuda") - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(), mk(), mk(), mk(0.4)
    O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    for _ in range(12):
        fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L)
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, causal=causal, dQ=dQ, dK=dK, dV=dV, workspace=ws)
        if two:
            fa.flash_attention_2_backward(Q, K, V, O, L, dO, None, causal=causal, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=6)
    torch.cuda.synchronize()
run(4, 16, 8192, 128, True)
run(4, 16, 8192, 64, False, two=True)
