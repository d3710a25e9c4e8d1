#!/usr/bin/env python3
"""What folding scale * log2(e) into the Q fragments would cost in accuracy (CPU, numpy, fp64).

The bf16 forward spends one v_fma_f32 per S element on s * c2 - mb (c2 = scale * log2 e).  The fold: Q' = bf16(c2 * Q) once per
row block, -mb as the MFMA chain's start value, p = exp2(S') directly.  Q is then rounded to bf16 a SECOND time (relative
2^-9 per element); everything else is unchanged.  This script measures, in fp64 on the bf16-rounded inputs the tests feed the
oracle, what that second rounding alone does to L and O -- against the gates of the parity tests (|dL| <= 1e-4 on ordinary
data, 1e-3 on the large-score cases; rel-L2(O) <= 5e-3, of which the kernel's own bf16 P already uses 2.1e-3)."""
import numpy as np


def bf16(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def attn(Qs, K, V):
    """Qs already carries the scale in log2 units: S2 = Qs K^T; returns (L natural, O)."""
    S2 = Qs.astype(np.float64) @ K.astype(np.float64).T
    m = S2.max(axis=1, keepdims=True)
    P = np.exp2(S2 - m)
    l = P.sum(axis=1, keepdims=True)
    return (m[:, 0] + np.log2(l[:, 0])) * np.log(2.0), (P / l) @ V.astype(np.float64)


def case(name, Q, K, V, scale, gate_L):
    c2 = np.float32(scale * np.log2(np.e))
    Q, K, V = bf16(Q), bf16(K), bf16(V)
    L0, O0 = attn(Q.astype(np.float64) * np.float64(c2), K, V)           # what the kernel computes today (fp32 fma: exact to 1e-7)
    L1, O1 = attn(bf16(Q * c2), K, V)                                     # the fold
    dL = np.abs(L1 - L0).max()
    rO = np.linalg.norm(O1 - O0) / np.linalg.norm(O0)
    verdict = "inside" if dL <= gate_L and rO <= 2.9e-3 else "BREAKS"
    print(f"{name:58s} max|L| {np.abs(L0).max():7.2f}  max|dL| {dL:9.2e} (gate {gate_L:.0e})  rel-L2(dO) {rO:9.2e}  -> {verdict}")


def main():
    rng = np.random.default_rng(0)
    u = lambda *s: rng.uniform(-0.5, 0.5, s).astype(np.float32)
    case("configs[0] (1,2,128,64) uniform(-0.5,0.5)", u(128, 64), u(128, 64), u(128, 64), 64 ** -0.5, 1e-4)
    case("configs[1] one head (4096,64) uniform(-0.5,0.5)", u(4096, 64), u(4096, 64), u(4096, 64), 64 ** -0.5, 1e-4)
    case("configs[2] one head (8192,128) uniform(-0.5,0.5)", u(2048, 128), u(8192, 128), u(8192, 128), 128 ** -0.5, 1e-4)
    n = lambda *s: rng.standard_normal(s).astype(np.float32)
    case("N(0,1) Q, K: scaled scores ~ N(0,1)  (2048 x 8192, d=128)", n(2048, 128), n(8192, 128), n(8192, 128), 128 ** -0.5, 1e-4)
    case("N(0,1) x 2: scaled scores ~ N(0,4)", 2 * n(2048, 128), n(8192, 128) * 2, n(8192, 128), 128 ** -0.5, 1e-4)
    for spike, gate in ((28.0, 1e-3), (87.0, 1e-3), (130.0, 1e-3)):
        Q, K, V = u(1024, 128), u(4096, 128), u(4096, 128)
        q = bf16(Q[700])
        K[2500] = q * (spike / (128 ** -0.5 * float(q @ q)))
        case(f"tests' spike construction: one key {spike:.0f} units above row 700", Q, K, V, 128 ** -0.5, gate)


if __name__ == "__main__":
    main()
