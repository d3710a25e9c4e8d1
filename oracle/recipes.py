"""Input recipes of the reference's test mains (harness helpers).

TEST INFRASTRUCTURE.  The randomised recipes call glibc's srand()/rand() through ctypes so
that the draws are the very numbers the reference's mains see (RAND_MAX = 2^31-1).
"""
import ctypes

import numpy as np

_libc = ctypes.CDLL("libc.so.6")
_libc.rand.restype = ctypes.c_int


def _draws(seed, n):
    _libc.srand(ctypes.c_uint(seed))
    out = np.empty(n, dtype=np.int64)
    r = _libc.rand
    for i in range(n):
        out[i] = r()
    return out


def fwd_rand(N=512, d=64, seed=42):
    """02_flash_attention_v2_forward/main.cu:28-33: per element i, Q[i], K[i], V[i] are drawn
    in that order as (rand() % 1000) / 1000.0f - 0.5f."""
    r = _draws(seed, 3 * N * d).reshape(N * d, 3)
    x = (r % 1000).astype(np.float32) / np.float32(1000.0) - np.float32(0.5)
    Q, K, V = (np.ascontiguousarray(x[:, t]).reshape(N, d) for t in range(3))
    return Q, K, V


def bwd_rand(N=128, d=64, seed=42, heads=1):
    """02_flash_attention_v2_backward/main.cu:221-227: per element, Q,K,V =
    ((rand()%2000)/1000.0f - 1.0f)*0.5f then dO = (...)*0.2f, interleaved.  heads > 1
    continues the same stream head-major (BASELINE config 1: B=1,H=2,N=128,d=64)."""
    r = _draws(seed, 4 * heads * N * d).reshape(heads, N * d, 4)
    u = (r % 2000).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)
    Q, K, V = (np.ascontiguousarray(u[:, :, t] * np.float32(0.5)).reshape(heads, N, d) for t in range(3))
    dO = np.ascontiguousarray(u[:, :, 3] * np.float32(0.2)).reshape(heads, N, d)
    if heads == 1:
        return Q[0], K[0], V[0], dO[0]
    return Q, K, V, dO


def fwd_simple():
    """02_flash_attention_v2_forward/main.cu:134-155 (N=d=4, scale=1)."""
    Q = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [1, 1, 0, 0]], dtype=np.float32)
    V = np.arange(1, 17, dtype=np.float32).reshape(4, 4)
    return Q, Q.copy(), V


def bwd_simple():
    """02_flash_attention_v2_backward/main.cu:78-107 (N=d=4, scale=1)."""
    eye = np.eye(4, dtype=np.float32)
    V = np.repeat(np.arange(1, 5, dtype=np.float32)[:, None], 4, axis=1)
    return eye, eye.copy(), V, eye.copy()


def naive00():
    """00_naive_attention/main.cpp:45-61: inputs and the four expected outputs."""
    Q = np.eye(2, dtype=np.float32)
    V = np.array([[1, 2], [3, 4]], dtype=np.float32)
    expected = np.array([[1.6604769, 2.6604770], [2.3395231, 3.3395231]], dtype=np.float32)
    return Q, Q.copy(), V, expected


def fa1_cases():
    """The literal forward cases of the superseded FA1 teaching step (01_flash_attention_v1/main.cu:195-345),
    judged there against naive_attention (scale 1/sqrt(d), main.cpp:8-38) with |diff| <= 1e-3 (:157-163): extra
    forward known-answer inputs.  Yields (name, Q, K, V) as fp32 [N, d]."""
    f = lambda rows: np.array(rows, dtype=np.float32)
    eye4 = np.eye(4, dtype=np.float32)
    yield "simple_2x4", f([[1, 0, 1, 0], [0, 1, 0, 1]]), f([[1, 0, 1, 0], [0, 1, 0, 1]]), f([[10, 20, 30, 40], [50, 60, 70, 80]])
    yield "identity_4x4", eye4, eye4, np.diag(np.arange(1, 5)).astype(np.float32)
    yield "uniform_3x2", np.ones((3, 2), np.float32), np.ones((3, 2), np.float32), f([[1, 2], [3, 4], [5, 6]])
    yield "orthogonal_2x2", f([[1, 0], [0, 1]]), f([[0, 1], [-1, 0]]), f([[10, 20], [30, 40]])
    yield "single_element", f([[1]]), f([[1]]), f([[42]])
    N, d = 8, 4
    Q = np.zeros((N, d), np.float32)
    Q[np.arange(d), np.arange(d)] = 1.0                        # (i == j) over an 8 x 4 matrix
    yield "diagonal_8x4", Q, Q.copy(), (np.arange(N)[:, None] * 10 + np.arange(d)[None, :]).astype(np.float32)
    N, d = 64, 32
    x = _draws(42, 3 * N * d).reshape(N * d, 3).astype(np.float32)       # (float)rand()
    rm = np.float32(2 ** 31 - 1)                                         # RAND_MAX converted to float
    yield ("random_64x32", (x[:, 0] / rm).reshape(N, d), (x[:, 1] / rm).reshape(N, d),
           (x[:, 2] / rm * np.float32(100)).reshape(N, d))
    yield "tiles_4x4", eye4, eye4, np.arange(1, 17, dtype=np.float32).reshape(4, 4)


def ring_pattern(N=5096, d=64):
    """util/attention_helper.h:151-173 (create_simple_test_data): Q = K = delta(i, j),
    V[i][j] = 4 i + j + 1."""
    Q = np.zeros((N, d), dtype=np.float32)
    idx = np.arange(min(N, d))
    Q[idx, idx] = 1.0
    V = (4.0 * np.arange(N, dtype=np.float32)[:, None] + np.arange(d, dtype=np.float32)[None, :] + 1.0)
    return Q, Q.copy(), V.astype(np.float32)


def ring_pattern_expected(N=5096, d=64):
    """Closed form of softmax(QK^T)V for ring_pattern at scale 1: rows
    r >= d attend uniformly; rows r < d weigh key r by e and every other key by 1."""
    _, _, V = ring_pattern(N, d)
    V64 = V.astype(np.float64)
    col = V64.sum(axis=0)
    O = np.tile(col / N, (N, 1))
    e = np.e
    for r in range(min(N, d)):
        O[r] = (col + (e - 1.0) * V64[r]) / (N + e - 1.0)
    return O


def compare_outputs(ref, test, rtol=1e-3, atol=1.0):
    """util/attention_helper.h:174-208: an element is wrong only if BOTH its relative
    error exceeds rtol AND its absolute error exceeds atol.  Returns the wrong count."""
    ref = np.asarray(ref, dtype=np.float32).ravel()
    test = np.asarray(test, dtype=np.float32).ravel()
    diff = np.abs(ref - test)
    rel = diff / (np.abs(ref) + np.float32(1e-8))
    return int(np.count_nonzero((rel > rtol) & (diff > atol)))
