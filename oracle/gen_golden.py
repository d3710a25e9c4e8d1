#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own code (oracle/_ref/libref_naive.so,
built by oracle/Makefile from /root/reference/src/util/naive_attention.h and
src/00_naive_attention/main.cpp).  Runs only in the dev container; the fixtures (data:
inputs + the reference's outputs) are committed, the reference itself never travels.

    python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle import recipes  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    R = oracle.ref()
    assert oracle.ref_self_test() == 0, "reference 00_naive_attention self test failed"

    # K0: 00_naive_attention/main.cpp:45-61
    Q, K, V, expected = recipes.naive00()
    O = oracle.naive_attention(Q, K, V, lib=R)
    np.savez(os.path.join(OUT, "k0_naive00.npz"), Q=Q, K=K, V=V, O=O, expected=expected)

    # K1: 02_flash_attention_v2_forward/main.cu:134-155, scale 1
    Q, K, V = recipes.fwd_simple()
    O, L = oracle.naive_forward_pass(Q, K, V, 1.0, lib=R)
    np.savez(os.path.join(OUT, "k1_fwd_simple.npz"), Q=Q, K=K, V=V, scale=np.float32(1.0), O=O, L=L)

    # K2: 02_flash_attention_v2_backward/main.cu:78-107, scale 1
    Q, K, V, dO = recipes.bwd_simple()
    O, L = oracle.naive_forward_pass(Q, K, V, 1.0, lib=R)
    dQ, dK, dV = oracle.naive_attention_backward(Q, K, V, dO, 1.0, lib=R)
    np.savez(os.path.join(OUT, "k2_bwd_simple.npz"), Q=Q, K=K, V=V, dO=dO, scale=np.float32(1.0),
             O=O, L=L, dQ=dQ, dK=dK, dV=dV)

    # K3: 02_flash_attention_v2_forward/main.cu:14-33 (N=512, d=64, srand(42))
    Q, K, V = recipes.fwd_rand(512, 64, 42)
    s = np.float32(1.0) / np.sqrt(np.float32(64))
    O, L = oracle.naive_forward_pass(Q, K, V, float(s), lib=R)
    np.savez_compressed(os.path.join(OUT, "k3_fwd_rand.npz"), Q=Q, K=K, V=V, scale=s, O=O, L=L)

    # K4: 02_flash_attention_v2_backward/main.cu:200-227 (N=128, d=64, srand(42))
    Q, K, V, dO = recipes.bwd_rand(128, 64, 42)
    O, L = oracle.naive_forward_pass(Q, K, V, float(s), lib=R)
    dQ, dK, dV = oracle.naive_attention_backward(Q, K, V, dO, float(s), lib=R)
    np.savez_compressed(os.path.join(OUT, "k4_bwd_rand.npz"), Q=Q, K=K, V=V, dO=dO, scale=s,
                        O=O, L=L, dQ=dQ, dK=dK, dV=dV)

    # cfg1: BASELINE config 1 (B=1,H=2,N=128,d=64): the K4 stream continued head-major,
    # each head run through the reference as an independent [N,d] problem.
    Q, K, V, dO = recipes.bwd_rand(128, 64, 42, heads=2)
    outs = {k: [] for k in ("O", "L", "dQ", "dK", "dV")}
    for h in range(2):
        O, L = oracle.naive_forward_pass(Q[h], K[h], V[h], float(s), lib=R)
        dQ, dK, dV = oracle.naive_attention_backward(Q[h], K[h], V[h], dO[h], float(s), lib=R)
        for k, v in zip(("O", "L", "dQ", "dK", "dV"), (O, L, dQ, dK, dV)):
            outs[k].append(v)
    np.savez_compressed(os.path.join(OUT, "cfg1_b1h2n128d64.npz"),
                        Q=Q[None], K=K[None], V=V[None], dO=dO[None], scale=s,
                        **{k: np.stack(v)[None] for k, v in outs.items()})

    # K5: 03_flash_attention_v2_ring/04_ring_attention.cu:19-21 with create_simple_test_data
    # (N=5096, d=64, scale 1).  Inputs are a closed-form pattern (recipes.ring_pattern), so
    # only a subset of the reference's output rows is stored.
    N, d = 5096, 64
    Q, K, V = recipes.ring_pattern(N, d)
    O, _ = oracle.naive_forward_pass(Q, K, V, 1.0, lib=R)
    rows = np.unique(np.concatenate([np.arange(0, 96), np.arange(2500, 2596), np.arange(N - 64, N),
                                     np.arange(0, N, 97)]))
    np.savez_compressed(os.path.join(OUT, "k5_ring_pattern_rows.npz"), N=np.int32(N), d=np.int32(d),
                        scale=np.float32(1.0), rows=rows.astype(np.int32), O_rows=O[rows])

    # K6: the literal forward cases of 01_flash_attention_v1/main.cu:195-345, expected outputs from the
    # reference's naive_attention (00_naive_attention/main.cpp:8-38), the checker that main uses (:155)
    fa1 = {}
    for name, Q, K, V in recipes.fa1_cases():
        fa1[name + "_Q"], fa1[name + "_K"], fa1[name + "_V"] = Q, K, V
        fa1[name + "_O"] = oracle.naive_attention(Q, K, V, lib=R)
    np.savez_compressed(os.path.join(OUT, "k6_fa1_cases.npz"), **fa1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
