"""ctypes doorway onto oracle/_build/liboracle.so (our C restatement) and, when it was
built in the dev container, oracle/_ref/libref_naive.so (the reference's own code).

TEST INFRASTRUCTURE.  numpy in / numpy out, float32, C-contiguous.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_naive.so")

_fp = ctypes.POINTER(ctypes.c_float)
_c_int = ctypes.c_int
_c_float = ctypes.c_float


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_ORACLE_SO) or \
            os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "naive_attention.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def _oracle():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_ORACLE_SO)
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def _reflib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError("oracle/_ref/libref_naive.so not built (needs /root/reference)")
        _ref = ctypes.CDLL(_REF_SO)
    return _ref


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else None


# ---------------------------------------------------------------- family 1 (verbatim order)
def naive_attention(Q, K, V, lib=None):
    """src/00_naive_attention/main.cpp:8-38 (scale fixed to 1/sqrt(d))."""
    Q, K, V = _f32(Q), _f32(K), _f32(V)
    N, d = Q.shape
    O = np.empty_like(Q)
    f = (lib or _oracle())
    fn = f.ref_naive_attention if lib is not None else f.oracle_naive_attention
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(O), _c_int(N), _c_int(d))
    return O


def naive_forward_pass(Q, K, V, scale=0.0, lib=None):
    """src/util/naive_attention.h:7-61 -> (O [N,d], L [N])."""
    Q, K, V = _f32(Q), _f32(K), _f32(V)
    N, d = Q.shape
    O = np.empty_like(Q)
    L = np.empty(N, dtype=np.float32)
    f = (lib or _oracle())
    fn = f.ref_naive_forward_pass if lib is not None else f.oracle_naive_forward_pass
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(O), _p(L), _c_int(N), _c_int(d), _c_float(scale))
    return O, L


def naive_attention_backward(Q, K, V, dO, scale, lib=None):
    """src/util/naive_attention.h:84-161 (O(N^3) Jacobian form) -> (dQ, dK, dV)."""
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    N, d = Q.shape
    dQ, dK, dV = np.empty_like(Q), np.empty_like(Q), np.empty_like(Q)
    dummyO, dummyL = np.zeros_like(Q), np.zeros(N, dtype=np.float32)
    f = (lib or _oracle())
    fn = f.ref_naive_attention_backward if lib is not None else f.oracle_naive_attention_backward
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(dummyO), _p(dummyL), _p(dO), _p(dQ), _p(dK), _p(dV),
       _c_int(N), _c_int(d), _c_float(scale))
    return dQ, dK, dV


def ref():
    """Handle that routes the three functions above to the REFERENCE's own build."""
    return _reflib()


def ref_self_test():
    """Runs the reference's own 2x2 known-answer main (00/main.cpp:40-85); 0 = pass."""
    f = _reflib().ref_naive00_self_test
    f.restype = _c_int
    return int(f())


# ---------------------------------------------------------------- family 2 (scalable, f64 accumulation)
def _as_slabs(x):
    x = _f32(x)
    lead = x.shape[:-2]
    return x.reshape((-1,) + x.shape[-2:]), lead


def attention_forward(Q, K, V, scale=0.0, causal=False, rows=None, heads=None):
    """O, L for [..., N, d] tensors (each leading index an independent head slab).

    rows=(row0, stride) restricts the work to a strided set of query rows and
    heads=(bh0, bh1) to a slab range; rows/slabs not computed are returned as NaN."""
    Qs, lead = _as_slabs(Q)
    Ks, _ = _as_slabs(K)
    Vs, _ = _as_slabs(V)
    BH, N, d = Qs.shape
    O = np.full_like(Qs, np.nan)
    L = np.full((BH, N), np.nan, dtype=np.float32)
    row0, stride = rows if rows is not None else (0, 1)
    bh0, bh1 = heads if heads is not None else (0, BH)
    fn = _oracle().oracle_attention_forward_rows_f64
    fn.restype = None
    fn(_p(Qs), _p(Ks), _p(Vs), _p(O), _p(L), _c_int(BH), _c_int(N), _c_int(d),
       _c_float(scale), _c_int(1 if causal else 0), _c_int(bh0), _c_int(bh1),
       _c_int(row0), _c_int(stride))
    return O.reshape(lead + (N, d)), L.reshape(lead + (N,))


def attention_backward(Q, K, V, dO, scale=0.0, causal=False):
    """dQ, dK, dV by the O(N^2 d) form dS = P o (dP - D); [..., N, d]."""
    Qs, lead = _as_slabs(Q)
    Ks, _ = _as_slabs(K)
    Vs, _ = _as_slabs(V)
    Gs, _ = _as_slabs(dO)
    BH, N, d = Qs.shape
    dQ, dK, dV = np.empty_like(Qs), np.empty_like(Qs), np.empty_like(Qs)
    fn = _oracle().oracle_attention_backward_f64
    fn.restype = None
    fn(_p(Qs), _p(Ks), _p(Vs), _p(Gs), _p(dQ), _p(dK), _p(dV), _c_int(BH), _c_int(N),
       _c_int(d), _c_float(scale), _c_int(1 if causal else 0))
    shp = lead + (N, d)
    return dQ.reshape(shp), dK.reshape(shp), dV.reshape(shp)


def attention_backward_head(Q, K, V, dO, scale=0.0, causal=False):
    """dQ, dK, dV of ONE head [N, d]: attention_backward's arithmetic with the query rows split over the OpenMP threads
    (a whole head at N = 8192 .. 65536 in seconds): what the at-size GPU tests compare a head of the backward with."""
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    assert Q.ndim == 2
    N, d = Q.shape
    dQ, dK, dV = np.empty_like(Q), np.empty_like(Q), np.empty_like(Q)
    fn = _oracle().oracle_attention_backward_head_f64
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(dO), _p(dQ), _p(dK), _p(dV), _c_int(N), _c_int(d), _c_float(scale),
       _c_int(1 if causal else 0))
    return dQ, dK, dV


# ---------------------------------------------------------------- family 3 (ring step)
def ring_step(Q, K, V, O, L, M, scale, last):
    """Folds one K/V shard into the (O, L, M) state in place (ring_attention_kernel.cu:67-137)."""
    for a in (Q, K, V, O, L, M):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    nq, d = Q.shape
    nk = K.shape[0]
    fn = _oracle().oracle_ring_step
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(O), _p(L), _p(M), _c_int(nq), _c_int(nk), _c_int(d),
       _c_float(scale), _c_int(1 if last else 0))


def set_threads(n):
    """OpenMP thread count for family 2 (cpu_baseline reports the count it used)."""
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def fwdbwd_rows(Q, K, V, dO, scale=0.0, rows=(0, 1)):
    """cpu_baseline sample: forward + backward share of a strided row set of one [N,d] head
    (14 N d flops per row).  Returns (O_rows, dQ_rows, dK_partial, dV_partial)."""
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    N, d = Q.shape
    row0, stride = rows
    n = len(range(row0, N, stride))
    O = np.empty((n, d), np.float32)
    dQ = np.empty((n, d), np.float32)
    dK = np.empty_like(Q)
    dV = np.empty_like(Q)
    fn = _oracle().oracle_fwdbwd_rows_f32
    fn.restype = None
    fn(_p(Q), _p(K), _p(V), _p(dO), _p(O), _p(dQ), _p(dK), _p(dV), _c_int(N), _c_int(d),
       _c_float(scale), _c_int(row0), _c_int(stride))
    return O, dQ, dK, dV


RB = 16      # ORACLE_RB in naive_attention.c


def fwdbwd_heads(Q, K, V, dO, scale=0.0, nblk=1, threads=1):
    """cpu_baseline: the naive fp32 forward + backward, ONE HEAD PER THREAD (thread t owns head t % BH of the [BH, N, d]
    inputs and `nblk` blocks of 16 consecutive query rows spread over it; nothing shared between threads).
    Returns (rows processed, O_rows [threads, nblk*16, d], dQ_rows, dK [threads, N, d], dV) -- 14 N d flops per row."""
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    BH, N, d = Q.shape
    nblk = max(1, min(int(nblk), N // RB))
    O = np.empty((threads, nblk * RB, d), np.float32)
    dQ = np.empty_like(O)
    dK = np.empty((threads, N, d), np.float32)
    dV = np.empty_like(dK)
    fn = _oracle().oracle_fwdbwd_heads_f32
    fn.restype = ctypes.c_long
    rows = fn(_p(Q), _p(K), _p(V), _p(dO), _p(O), _p(dQ), _p(dK), _p(dV), _c_int(BH), _c_int(N), _c_int(d),
              _c_float(scale), _c_int(nblk), _c_int(threads))
    return int(rows), O, dQ, dK, dV


def fwdbwd_heads_rows(N, nblk):
    """The query rows fwdbwd_heads works on in each head (same spacing rule as the C code)."""
    nblk = max(1, min(int(nblk), N // RB))
    return np.concatenate([np.arange(RB) + (b * (N // RB) // nblk) * RB for b in range(nblk)])


def get_threads():
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_get_max_threads.restype = ctypes.c_int
        return int(omp.omp_get_max_threads())
    except OSError:
        return 1
