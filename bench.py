#!/usr/bin/env python3
"""bench.py -- FA2 fwd+bwd TFLOP/s at (B=4,H=16,N=8192,d=128) on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic bf16 Q/K/V/dO already
resident in HBM: fa2_forward then fa2_backward (delta + the single five-product kernel + dQ output pass) through the C ABI
of libfa2_mi355x.so.  W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier +
torch.cuda.synchronize() on both sides; rank 0 prints ONE JSON line.

N > 1 ranks: the path shards over independent (batch, head) slabs with no data-path
collective, so every rank runs the full (4,16,8192,128) batch of its own ("weak" scaling) and
`value` is the total flops of all ranks / the slowest rank's time.  The sequence-sharded ring
forward (the path's one real exchange step, RCCL send/recv over xGMI) is timed afterwards and
reported in the extra "ring" object when the ring library is available.

Flop model: fwd 4 B H N^2 d, bwd 10 B H N^2 d (five block products), fwd+bwd 14.
At this shape fa2_backward runs the single five-product kernel (csrc/fa2_bwd_fused.hip); the two-kernel
form, which executes seven products for the same five, is timed beside it (`two_kernel_backward_ms`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # MI355X dense bf16 MFMA: 256 CU x 4096 flop/clk x 2.4 GHz
# What the chip sustains when every SIMD issues nothing but v_mfma_f32_32x32x16_bf16 on random
# operands (tools/probes/mfma_power.hip): the clock settles near 1.77 GHz.  Reported beside the
# nominal peak; `frac` is always against the nominal one.
SUSTAINED_MFMA_TFLOPS = 1840.0
# HBM bytes per launch come from the rocprofv3 --pmc passes (separate runs, MI355X_MICROARCH.md: FETCH_SIZE doubled on
# gfx950, plus WRITE_SIZE), summarised by tools/pmc_summary.py into this file; the JSON line names it.  A kernel that
# is not in the file (the profile predates it) reports traffic null rather than a stale constant.
PMC_TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")
PEAK_FP8_TFLOPS = 5033.2    # dense fp8 MFMA (v_mfma_f32_32x32x64_f8f6f4): twice the bf16 rate
# PMC evidence for the two side configurations (MFMA-busy, clock, VALU / LDS / wait split, FETCH / WRITE): see profiles/README.md
SIDE_PMC_FILE = os.path.join("profiles", "r3_e_side_pmc_summary.txt")
B, H, N, D = 4, 16, 8192, 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ring", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--sustained-steps", type=int, default=350)     # 350 x ~6 ms > 2 s
    return ap.parse_args()


def timed(fn, iters, torch):
    """Average milliseconds per call, HIP events on the stream the kernels are launched on
    (torch's current stream is what the C ABI receives)."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def pmc_traffic(kernel):
    try:
        with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as f:
            tab = json.load(f)
        ent = tab["kernels"].get(kernel)
        return (ent["bytes_per_launch"], tab.get("source")) if ent else (None, tab.get("source"))
    except (OSError, ValueError, KeyError):
        return None, None


def median_ms(fn, torch, repeats=3, iters=10):
    """Median over `repeats` timings of `iters` launches each (HIP events on the launch stream)."""
    vals = sorted(timed(fn, iters, torch) for _ in range(repeats))
    return vals[len(vals) // 2]


def cpu_baseline(torch):
    """The reference's naive CPU attention (oracle/naive_attention.c: oracle_fwdbwd_heads_f32, the triple loops of
    src/util/naive_attention.h:7-61 and the O(N^2 d) form of :84-161 in its own fp32 arithmetic) timed on this host the
    way SURVEY 8d prescribes: ONE HEAD PER THREAD, nothing shared between threads (each keeps the dK / dV sums of its own
    head in its own memory; no locks, no merging), every thread working on blocks of 16 consecutive query rows of its head.
    A BOUNDED sample of the bench workload: the block count is chosen from a one-block calibration run so that the timed
    run takes about 12 s."""
    import numpy as np
    import oracle
    cores = oracle.get_threads()
    heads = min(B * H, cores)
    rng = np.random.default_rng(0)
    Q, K, V = (rng.uniform(-0.5, 0.5, (heads, N, D)).astype(np.float32) for _ in range(3))
    dO = rng.uniform(-0.2, 0.2, (heads, N, D)).astype(np.float32)
    t0 = time.perf_counter()
    rows, *_ = oracle.fwdbwd_heads(Q, K, V, dO, 0.0, nblk=1, threads=cores)      # calibration (also pages everything in)
    per_block = time.perf_counter() - t0
    nblk = max(1, min(N // oracle.RB, int(12.0 / max(per_block, 1e-3))))
    t0 = time.perf_counter()
    rows, *_ = oracle.fwdbwd_heads(Q, K, V, dO, 0.0, nblk=nblk, threads=cores)
    dt = time.perf_counter() - t0
    flops = 14.0 * N * D * rows
    return {"value": flops / dt / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "gflops_per_core": round(flops / dt / 1e9 / cores, 3),
            "sample": f"naive fp32 fwd+bwd, one head per thread: {cores} threads x {nblk} blocks of {oracle.RB} query rows "
                      f"= {rows} of the workload's {B * H * N} rows (N={N}, d={D}; {flops / 1e9:.0f} GFLOP in {dt:.2f} s)"}


def main():
    args = parse()
    import torch
    import cuda_flashattention_amd as fa

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local % torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("FA2_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path on one GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    mk = lambda s: ((torch.rand(B, H, N, D, device=dev, generator=g) - 0.5) * s).to(torch.bfloat16)
    Q, K, V, dO = mk(1.0), mk(1.0), mk(1.0), mk(0.4)   # the reference's value ranges (main.cu:30-32, :226)
    O = torch.empty_like(Q)
    L = torch.empty(B, H, N, dtype=torch.float32, device=dev)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, D, 0), dtype=torch.uint8, device=dev)
    scale = 1.0 / D ** 0.5

    def fwd():
        fa.flash_attention_2_forward(Q, K, V, scale, O=O, L=L)

    def bwd(phases=7):
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, dQ=dQ, dK=dK, dV=dV, workspace=ws, phases=phases)

    def step():
        fwd()
        bwd()

    # ---- per-kernel launch durations (HIP events, same stream), for the roofline object.  Taken BEFORE the
    # timed region: besides being needed below, these ~40 launches bring the GPU out of its idle power state,
    # so that the W warm-up steps and the K timed steps that follow run at the sustained clock whatever W is
    # (a cold start costs ~5 % on the first few steps).
    for _ in range(4):          # one-time work (code-object load, LDS limits) and the clock ramp stay out of every timing below
        step()
    torch.cuda.synchronize()
    it = max(5, min(args.steps, 10))
    k_ms = {
        "fa2_fwd1_bf16_kernel": median_ms(fwd, torch, 3, it),
        "fa2_bwd_delta_kernel": median_ms(lambda: bwd(1), torch, 3, it),
        # what fa2_backward runs at this shape: the single five-product kernel (with its control-block memset and the
        # fp32 -> bf16 output pass of dQ: ~0.08 ms of the figure)
        "fa2_bwd_fused_kernel": median_ms(lambda: bwd(8), torch, 3, it),
    }
    # the two-kernel form of the same backward (seven products), for comparison: FA2_BACKWARD_PATH=two_kernel selects it
    two_kernel_ms = {"fa2_bwd_dq_kernel": median_ms(lambda: bwd(2), torch, 3, it),
                     "fa2_bwd_dkdv_kernel": median_ms(lambda: bwd(4), torch, 3, it)}
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sustained: the same step for >= 2 s back to back (the kernels are power-limited: a 0.1 s window after a few dozen
    # launches does not show what the chip holds), with the mean shader clock over the window from the device's own
    # counters (fa2_read_clocks: s_memtime / s_memrealtime x 100 MHz)
    sustained = None
    if rank == 0 and not args.no_sustained:
        n_sus = max(args.sustained_steps, 1)
        torch.cuda.synchronize()
        clk0 = fa.ops.read_clocks()
        t0s = time.perf_counter()
        for _ in range(n_sus):
            step()
        clk1 = fa.ops.read_clocks()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0s
        mhz = fa.ops.mean_shader_clock_mhz(clk0, clk1)
        sustained = {"steps": n_sus, "seconds": round(dts, 3), "ms_per_step": round(dts / n_sus * 1e3, 4),
                     "tflops": round(14.0 * B * H * N * N * D * n_sus / dts / 1e12, 2),
                     "mean_shader_clock_mhz": round(mhz, 1),
                     "clock_source": "fa2_read_clocks before/after the window: per XCC d(s_memtime) / d(s_memrealtime) x 100 MHz, mean over the XCCs"}

    # ---- what THIS device sustains on bare bf16 MFMAs with random operands (fa2_mfma_probe: every SIMD, nothing else): devices
    # of one pool differ by several per cent on power-limited kernels, so the line carries the box's own ceiling beside the
    # nominal peak.  After the timed region; ~0.2 s.
    box = None
    if rank == 0:
        try:
            tf, mhz = fa.ops.bare_mfma_tflops()
            box = {"bare_mfma_tflops": round(tf, 1), "clock_mhz": round(mhz, 1),
                   "what": "fa2_mfma_probe: back-to-back v_mfma_f32_32x32x16_bf16 on random operands, one wave per SIMD, ~0.15 s"}
        except Exception as e:
            box = {"error": repr(e)}

    flops_step = 14.0 * B * H * N * N * D
    ms_per_step = elapsed / args.steps * 1e3
    value = world * flops_step / (ms_per_step * 1e-3) / 1e12

    # MFMA flops each launch executes = the ALGORITHMIC flops it delivers: forward 2 block products, backward 5 (S, dP, dV,
    # dK, dQ, each formed once by the single kernel).  The two-kernel form executes 3 + 4 for the same five.
    prod = 2.0 * B * H * N * N * D
    k_flops = {"fa2_fwd1_bf16_kernel": 2 * prod, "fa2_bwd_fused_kernel": 5 * prod}
    k_alg = dict(k_flops)
    dom = max(k_flops, key=lambda k: k_ms[k])
    achieved = k_alg[dom] / (k_ms[dom] * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(dom)
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC)", "traffic_source": traffic_src and PMC_TRAFFIC_FILE,
                "sustained_mfma_peak": SUSTAINED_MFMA_TFLOPS, "frac_of_sustained": round(achieved / SUSTAINED_MFMA_TFLOPS, 4),
                "flops_per_launch": k_alg[dom], "ms_per_launch": round(k_ms[dom], 4),
                "kernels_ms": {k: round(v, 4) for k, v in k_ms.items()},
                "kernels_frac_executed": {k: round(k_flops[k] / (k_ms[k] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) for k in k_flops},
                "kernels_frac_algorithmic": {k: round(k_alg[k] / (k_ms[k] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) for k in k_alg},
                "two_kernel_backward_ms": {k: round(v, 4) for k, v in two_kernel_ms.items()},
                "whole_path_frac": round(value / world / PEAK_BF16_TFLOPS, 4)}
    if box is not None:
        roofline["this_device"] = box
        if "bare_mfma_tflops" in box:
            roofline["frac_of_this_device_bare_mfma"] = round(achieved / box["bare_mfma_tflops"], 4)
            roofline["whole_path_frac_of_this_device_bare_mfma"] = round(value / world / box["bare_mfma_tflops"], 4)

    out = {
        "metric": "FA2 fwd+bwd TFLOP/s at (B=4,H=16,N=8192,d=128); % MFMA peak",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "FA2 forward+backward bf16, (B=4,H=16,N=8192,d=128) per GPU, non-causal "
                               "(BASELINE configs[2]); flops = 14 B H N^2 d",
                   "B": B, "H": H, "N": N, "d": D, "parallelism": f"head-sharded replicas x{world}"},
        "pct_mfma_peak": round(100.0 * value / world / PEAK_BF16_TFLOPS, 2),
        "roofline": roofline,
    }
    if sustained is not None:
        out["sustained"] = sustained

    # ---- side figures SURVEY 8d asks for beside the headline (timed BEFORE the CPU baseline idles the GPU): forward only at BASELINE configs[1] (bf16)
    # and the fp8 causal forward of configs[4] (B and H unspecified there: B=1, H=16)
    if rank == 0:
        try:
            extra = {}
            def fwd_only(Bx, Hx, Nx, dx, dt, causal):
                mkx = lambda: (torch.rand(Bx, Hx, Nx, dx, device=dev) - 0.5).to(dt)
                q, k, v = mkx(), mkx(), mkx()
                o = torch.empty(Bx, Hx, Nx, dx, dtype=torch.bfloat16, device=dev)
                l = torch.empty(Bx, Hx, Nx, dtype=torch.float32, device=dev)
                # fp8: the caller owns the scratch (V transposed + key norms), as a serving loop would -- no hipMallocAsync /
                # hipFreeAsync inside the timed calls
                wsx = fa.ops.forward_fp8_workspace(Bx, Hx, Nx, dx, dev) if dt == torch.float8_e4m3fn else None
                f = lambda: fa.flash_attention_2_forward(q, k, v, None, causal=causal, O=o, L=l, workspace=wsx)
                for _ in range(50):         # code-object load and clock ramp stay out of the timings
                    f()
                ms = median_ms(f, torch, 3, 20)          # median of three runs of 20 launches
                fl = 4.0 * Bx * Hx * Nx * Nx * dx * (0.5 if causal else 1.0)
                peak = PEAK_FP8_TFLOPS if dt == torch.float8_e4m3fn else PEAK_BF16_TFLOPS
                return {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "timing": "median of 3 x 20 launches after 50 warm-up launches",
                        "roofline": {"bound": "mfma", "achieved": round(fl / ms / 1e9, 1), "peak": peak, "unit": "TFLOP/s",
                                     "frac": round(fl / ms / 1e9 / peak, 4), "counters": SIDE_PMC_FILE,
                                     "limiting_unit": "VALU issue under the chip's power limit: 4 (bf16) / 3.5 - 4 (fp8) VALU instructions per "
                                                      "S element against half (d = 64) or a quarter (fp8) of the MFMA cycles of bf16 d = 128; "
                                                      "45.1 % / 53.0 % MFMA-busy at 2.04 GHz in the PMC passes (DESIGN.md section 3)"}}
            extra["fwd_bf16_cfg2_(4,16,4096,64)"] = fwd_only(4, 16, 4096, 64, torch.bfloat16, False)
            extra["fwd_fp8_e4m3_causal_cfg5_(1,16,32768,128)"] = fwd_only(1, 16, 32768, 128, torch.float8_e4m3fn, True)
            # the headline shape with a causal mask (past the reference, which has none): forward + backward, 7 B H N^2 d flops
            def causal_step():
                fa.flash_attention_2_forward(Q, K, V, scale, causal=True, O=O, L=L)
                fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, causal=True, dQ=dQ, dK=dK, dV=dV, workspace=ws)
            for _ in range(10):
                causal_step()
            msc = median_ms(causal_step, torch, 3, 10)
            extra["fwd_bwd_bf16_causal_(4,16,8192,128)"] = {"ms": round(msc, 4), "tflops": round(7.0 * B * H * N * N * D / msc / 1e9, 1),
                                                            "timing": "median of 3 x 10 steps after 10 warm-up steps"}
            # the backward at head_dim 64 (round 4: the single kernel built for d = 64) beside the two-kernel form it replaced
            def bwd64():
                Bx, Hx, Nx, dx = 4, 16, 8192, 64
                mkx = lambda s: ((torch.rand(Bx, Hx, Nx, dx, device=dev) - 0.5) * s).to(torch.bfloat16)
                q, k, v, g_ = mkx(1.0), mkx(1.0), mkx(1.0), mkx(0.4)
                o, l = fa.flash_attention_2_forward(q, k, v)
                w = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(Bx, Hx, Nx, dx, 0), dtype=torch.uint8, device=dev)
                gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
                run = lambda ph: fa.flash_attention_2_backward(q, k, v, o, l, g_, dQ=gq, dK=gk, dV=gv, workspace=w, phases=ph)
                for _ in range(5):
                    run(7)
                one = median_ms(lambda: run(7), torch, 3, 10)
                two = median_ms(lambda: run(1), torch, 3, 10) + median_ms(lambda: run(2), torch, 3, 10) + median_ms(lambda: run(4), torch, 3, 10)
                fl = 10.0 * Bx * Hx * Nx * Nx * dx
                return {"ms": round(one, 4), "tflops": round(fl / one / 1e9, 1), "two_kernel_form_ms": round(two, 4),
                        "timing": "fa2_backward (delta + single kernel + output pass), median of 3 x 10 launches; flops = 10 B H N^2 d"}
            extra["bwd_bf16_d64_(4,16,8192,64)"] = bwd64()
            fwd()                                                    # leave O, L as the non-causal forward's
            # the reference's FlashAttention-1 step restated (scalar fp32, one head, no MFMA): a DIDACTIC row, not a target
            Nf, df = 4096, 64
            qf, kf, vf = (torch.rand(Nf, df, device=dev) - 0.5 for _ in range(3))
            of = torch.empty(Nf, df, device=dev)
            lf, mf = torch.empty(Nf, device=dev), torch.empty(Nf, device=dev)
            lib = fa._capi.lib()
            f1 = lambda: lib.flash_attention(qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), of.data_ptr(), lf.data_ptr(), mf.data_ptr(),
                                             Nf, df, 32, 1024)
            torch.cuda.synchronize()          # flash_attention runs on the default stream, like the reference's
            for _ in range(3):
                f1()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                f1()
            torch.cuda.synchronize()
            msf = (time.perf_counter() - t0) / 5 * 1e3
            extra["fa1_didactic_fp32_(1,1,4096,64)"] = {"ms": round(msf, 4), "tflops": round(4.0 * Nf * Nf * df / msf / 1e9, 3),
                                                        "note": "01_flash_attention_v1 restated: one thread per query row, fp32 FMAs"}
            out["side_figures"] = extra
        except Exception as e:
            out["side_figures"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(torch)
        except Exception as e:  # the baseline is a reported side figure, never fatal
            out["cpu_baseline"] = {"error": repr(e)}

    # The ring leg is a side figure: it must never take the headline line down with it.  It runs in a
    # worker thread under a deadline; if the transport wedges (an RCCL hang cannot be interrupted) the
    # line is printed without it and the process leaves through os._exit.
    hung = False
    if not args.no_ring:
        import threading
        box = {}

        def ring_leg():
            try:
                from cuda_flashattention_amd import ring
                torch.cuda.set_device(local % torch.cuda.device_count())
                box["ring"] = ring.bench_ring(dist, rank, world, steps=max(2, min(args.steps, 5)), warmup=1)
            except Exception as e:
                box["ring"] = {"error": repr(e)}

        # RCCL prints a version banner on the process's stdout when the environment sets NCCL_DEBUG=VERSION: while the leg runs,
        # file descriptor 1 points at stderr, so that stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            th = threading.Thread(target=ring_leg, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("FA2_BENCH_RING_TIMEOUT", "180")))
            hung = th.is_alive()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if hung:
            out["ring"] = {"error": "ring leg did not finish within its deadline; skipped"}
            out["ring_hang"] = True
        elif box.get("ring") is not None:
            out["ring"] = box["ring"]

    if rank == 0:
        print(json.dumps(out), flush=True)
    sys.stdout.flush()
    if hung:
        # A wedged transport cannot be interrupted (the worker thread sits inside RCCL): every rank leaves through os._exit.
        # The headline line is out and flushed and says so ("ring_hang": true), and rank 0 leaves with a NON-ZERO status so that
        # a harness that only reads exit codes sees the hang too.  Under torch.distributed.run a non-zero status of any rank
        # makes the launcher kill the others: the other ranks therefore sleep first (rank 0 is never the one killed early) and
        # then leave with 0.
        if rank != 0:
            time.sleep(5.0)
            os._exit(0)
        os._exit(3)
    if dist is not None and not args.no_ring:
        # After a ring leg the ranks may disagree on whether it finished (each has its own deadline): no
        # further collective, every rank simply leaves.  The timed region and its barriers are long past.
        if rank != 0:
            time.sleep(2.0)
        os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
