import sys, os, torch
sys.path.insert(0, os.getcwd())
import cuda_flashattention_amd as fa
def med(fn):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    v = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1) / 20)
    return sorted(v)[1]
for (B, H, N, d, causal) in ((1, 16, 32768, 128, True), (1, 16, 32768, 128, False), (4, 16, 8192, 128, False), (4,16,4096,128,True)):
    Q, K, V = ((torch.rand(B, H, N, d, device="cuda") - 0.5).to(torch.float8_e4m3fn) for _ in range(3))
    O = torch.empty(B, H, N, d, dtype=torch.bfloat16, device="cuda"); L = torch.empty(B, H, N, device="cuda")
    ms = med(lambda: fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L))
    fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
    print(f"{os.environ.get('FA2_LIB_PATH','new')}: fp8 fwd ({B},{H},{N},{d}) causal={causal}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
