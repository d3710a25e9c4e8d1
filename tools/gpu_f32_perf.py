"""Dev aid: the fp32 (reference-type) path's speed -- fa2_forward / fa2_backward with FA2_DTYPE_F32 (exact f32 MFMA kernels)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (B, H, N, d) in ((1, 1, 512, 64), (1, 1, 5096, 64), (4, 16, 4096, 64), (4, 16, 8192, 128)):
    Q, K, V, dO = (torch.rand(B, H, N, d, device="cuda") - 0.5 for _ in range(4))
    O, L = fa.flash_attention_2_forward(Q, K, V)
    tf = t(lambda: fa.flash_attention_2_forward(Q, K, V))
    tb = t(lambda: fa.flash_attention_2_backward(Q, K, V, O, L, dO))
    fl = 4.0 * B * H * N * N * d
    print(f"fp32 ({B},{H},{N},{d}): fwd {tf:.3f} ms {fl / tf / 1e9:.1f} TFLOP/s   bwd {tb:.3f} ms {2.5 * fl / tb / 1e9:.1f} TFLOP/s", flush=True)
