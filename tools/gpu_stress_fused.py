"""Stress run of the single-kernel backward: many launches over random eligible shapes (causal and not), each checked
bit for bit against its own first result and within bf16 rounding against the two-kernel form (dev aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
lib = fa._capi.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = time.time()
bad = 0
for it in range(rounds):
    B, H, N = int(rng.integers(1, 4)), int(rng.integers(1, 20)), 256 * int(rng.integers(1, 33))
    if rng.integers(0, 3) == 0:
        N = max(1, N - int(rng.integers(1, 200)))           # ragged: padded inside (or the two kernels, by the library's rule)
    while B * H * N > 4 * 16 * 8192:
        H = max(1, H // 2)
    causal = bool(rng.integers(0, 2))
    d = int(rng.choice([64, 128]))
    g = torch.Generator(device="cuda").manual_seed(1000 + it)
    mk = lambda s: ((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    O, L = fa.flash_attention_2_forward(Q, K, V, causal=causal)
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(6):
        o = [torch.empty_like(Q) for _ in range(3)]
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, causal=causal, dQ=o[0], dK=o[1], dV=o[2], workspace=ws)
        outs.append(o)
    two = [torch.empty_like(Q) for _ in range(3)]
    for ph in (1, 6):
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, causal=causal, dQ=two[0], dK=two[1], dV=two[2], workspace=ws, phases=ph)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(o, outs[0]))
    err = max(float((a.float() - b.float()).norm() / b.float().norm()) for a, b in zip(outs[0], two))
    ok = same and err < 1.5e-3
    bad += not ok
    plan = lib.fa2_backward_plan(B, H, N, d, 0, int(causal), None)
    print(f"{it:3d} B{B} H{H} N{N} d{d} causal={int(causal)} plan={plan}: repeatable={same} rel vs two-kernel={err:.2e}{'' if ok else '  <-- BAD'}", flush=True)
print(f"{rounds} shapes x 6 launches in {time.time() - t0:.0f} s, bad = {bad}")
sys.exit(1 if bad else 0)
