"""Host-side mirror of the reference's ring-attention interface
(ring_attention_forward, src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:143-156;
init_nccl_comm / ring_exchange_kv, src/util/nccl_utils.h:29-56, :115-142).

One process per GPU, launched by torch.distributed.run.  torch.distributed replaces the
reference's MPI as the *bootstrap* only (it carries the 128-byte RCCL unique id from rank 0);
the data path is libfa2_ring_mi355x.so: RCCL send/recv over xGMI on its own stream, event-fenced
against the FA2 step kernel (include/fa2_ring_mi355x.h).

The schedules themselves (relay, mesh, causal zig-zag, backward) live in ONE place,
csrc/ring/fa2_ring.cpp; this module only marshals tensors into the C ABI.  The library does all
device work through a backend table (fa2_ring_backend): RCCL + HIP + the kernels in the product,
and in tests/ a one-GPU loopback transport, a CPU simulator and a gloo transport that run the same
C++ schedule code at P = 2..8 (tests/test_gpu_ring_loopback.py, test_ring_sim.py, test_ring_gloo.py).
"""
import ctypes
import math
import os
import time

import torch

from . import _capi
from ._capi import FA2_DTYPE_BF16, FA2_DTYPE_F32, check

RELAY = 0
MESH = 1
_SCHEDULES = {"relay": RELAY, "mesh": MESH, RELAY: RELAY, MESH: MESH}

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# mirrors include/fa2_ring_mi355x.h declaration by declaration
RING_SIGNATURES = {
    "fa2_ring_get_unique_id": (_i, [_vp]),
    "fa2_ring_ctx_create": (_i, [ctypes.POINTER(_vp), _vp, _i, _i]),
    "fa2_ring_ctx_create_from_comm": (_i, [ctypes.POINTER(_vp), _vp, _i, _i]),
    "fa2_ring_default_backend": (_i, [_vp]),
    "fa2_ring_ctx_create_with_backend": (_i, [ctypes.POINTER(_vp), _vp, _i, _i]),
    "fa2_ring_ctx_destroy": (_i, [_vp]),
    "fa2_ring_ctx_set_reserved_cus": (_i, [_vp, _i]),
    "fa2_ring_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "fa2_ring_attention_forward": (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "fa2_ring_attention_forward_causal": (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "fa2_ring_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fa2_ring_backward_block_workspace": (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "fa2_ring_attention_backward": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "fa2_ring_attention_backward_causal": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ring_attention_forward": (_i, [_vp] * 5 + [_i, _i, _i, _f, _vp, _i, _i]),
    "fa2_ring_exchange_kv": (_i, [_vp] * 5 + [_sz, _vp]),
}

_ring = None


def ring_lib():
    """libfa2_ring_mi355x.so (needs torch imported first so that librccl.so.1 / libamdhip64.so.7
    resolve to the copies torch already loaded).  Raises if it is missing."""
    global _ring
    if _ring is None:
        _capi.lib()
        if not os.path.exists(_capi.RING_LIB_PATH):
            raise ImportError(f"{_capi.RING_LIB_PATH} is missing: `make -C cuda_flashattention_amd/csrc ring`")
        h = ctypes.CDLL(_capi.RING_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in RING_SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _ring = h
    return _ring


def _dtype_code(t):
    if t.dtype == torch.bfloat16:
        return FA2_DTYPE_BF16
    if t.dtype == torch.float32:
        return FA2_DTYPE_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def shard_rows(total_seq_len, rank, nranks):
    """Row range [lo, hi) of rank's contiguous shard; N must divide (04_ring_attention.cu:55-63)."""
    if total_seq_len % nranks != 0:
        raise ValueError("seq_len must be divisible by nranks!")
    n = total_seq_len // nranks
    return rank * n, (rank + 1) * n


def zigzag_rows(total_seq_len, rank, nranks):
    """Global row indices of rank's local rows under the causal ring's zig-zag sharding: chunk `rank` then
    chunk 2P-1-rank of the 2P chunks of N/(2P) rows (fa2_ring_attention_forward_causal)."""
    if total_seq_len % (2 * nranks) != 0:
        raise ValueError("seq_len must be divisible by 2 * nranks!")
    c = total_seq_len // (2 * nranks)
    a, b = rank, 2 * nranks - 1 - rank
    return list(range(a * c, (a + 1) * c)) + list(range(b * c, (b + 1) * c))


def kv_owner(rank, step, nranks):
    """Owner of the K/V shard rank computes on at `step` (ring_attention_kernel.cu:198)."""
    return (rank - step + nranks) % nranks


class RingContext:
    """RCCL communicator + comm stream + events for one rank (fa2_ring_ctx).  Collective."""

    def __init__(self, dist=None, rank=0, nranks=1):
        lib = ring_lib()
        self.rank, self.nranks = rank, nranks
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            check(lib.fa2_ring_get_unique_id(ident), "fa2_ring_get_unique_id")
        if nranks > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)     # the reference's MPI_Bcast (nccl_utils.h:42)
            ident = ctypes.create_string_buffer(box[0], 128)
        self._h = _vp()
        check(lib.fa2_ring_ctx_create(ctypes.byref(self._h), ident, rank, nranks), "fa2_ring_ctx_create")
        # CUs the ring backward's block kernels leave to the exchanges beside them: tunable per run, no rebuild
        if os.environ.get("FA2_RING_RESERVED_CUS"):
            check(lib.fa2_ring_ctx_set_reserved_cus(self._h, int(os.environ["FA2_RING_RESERVED_CUS"])), "fa2_ring_ctx_set_reserved_cus")
        self._ws = None

    def close(self):
        if getattr(self, "_h", None):
            ring_lib().fa2_ring_ctx_destroy(self._h)
            self._h = None

    def workspace(self, B, H, n_local, d, dtype_code, schedule, device):
        need = ring_lib().fa2_ring_workspace_bytes(B, H, n_local, d, dtype_code, self.nranks, schedule)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        return self._ws


def ring_attention_forward(ctx, Q_local, K_local, V_local, softmax_scale=None, schedule="relay",
                           O_local=None, L_local=None, stream=None, causal=False):
    """O_local, L_local for this rank's rows.  Mirrors ring_attention_forward(Q_local, K_local,
    V_local, O_local, L_local, total_seq_len, local_seq_len, head_dim, scale, comm, rank, nranks);
    tensors [B,H,N/P,d] (or [N/P,d]).  K_local / V_local are left intact.  causal=True: the local rows
    are the rank's two zig-zag chunks (zigzag_rows); bf16 only."""
    if Q_local.dim() == 2:
        B, H = 1, 1
        n, d = Q_local.shape
    else:
        B, H, n, d = Q_local.shape
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    sched = _SCHEDULES[schedule]
    code = _dtype_code(Q_local)
    if O_local is None:
        O_local = torch.empty_like(Q_local)
    if L_local is None:
        L_local = torch.empty(Q_local.shape[:-1], dtype=torch.float32, device=Q_local.device)
    ws = ctx.workspace(B, H, n, d, code, sched, Q_local.device)
    s = stream if stream is not None else torch.cuda.current_stream()
    fn = ring_lib().fa2_ring_attention_forward_causal if causal else ring_lib().fa2_ring_attention_forward
    st = fn(
        ctx._h, Q_local.data_ptr(), K_local.data_ptr(), V_local.data_ptr(), O_local.data_ptr(),
        L_local.data_ptr(), B, H, n * ctx.nranks, n, d, scale, code, sched,
        ws.data_ptr(), ws.numel(), s.cuda_stream)
    check(st, "fa2_ring_attention_forward")
    return O_local, L_local


def ring_attention_backward(ctx, Q_local, K_local, V_local, O_local, L_local, dO_local, softmax_scale=None, stream=None,
                            causal=False):
    """dQ, dK, dV for this rank's rows / keys (fa2_ring_attention_backward[_causal]): O_local, L_local are the ring
    forward's outputs.  bf16 [B,H,N/P,d]; causal=True: the zig-zag layout of the causal forward."""
    B, H, n, d = Q_local.shape
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    code = _dtype_code(Q_local)
    lib = ring_lib()
    need = lib.fa2_ring_backward_workspace_bytes(B, H, n, d, code, ctx.nranks)
    if getattr(ctx, "_bws", None) is None or ctx._bws.numel() < need:
        ctx._bws = torch.empty(max(need, 256), dtype=torch.uint8, device=Q_local.device)
    dQ, dK, dV = torch.empty_like(Q_local), torch.empty_like(K_local), torch.empty_like(V_local)
    s = stream if stream is not None else torch.cuda.current_stream()
    fn = lib.fa2_ring_attention_backward_causal if causal else lib.fa2_ring_attention_backward
    st = fn(ctx._h, Q_local.data_ptr(), K_local.data_ptr(), V_local.data_ptr(),
            O_local.data_ptr(), L_local.data_ptr(), dO_local.data_ptr(), dQ.data_ptr(),
            dK.data_ptr(), dV.data_ptr(), B, H, n * ctx.nranks, n, d, scale, code,
            ctx._bws.data_ptr(), ctx._bws.numel(), s.cuda_stream)
    check(st, "fa2_ring_attention_backward")
    return dQ, dK, dV


# ------------------------------------------------------------------------------------------
# bench.py leg: ring forward at N = 8192 * P, d = 128 (BASELINE configs[3] at P = 8)
# ------------------------------------------------------------------------------------------
def bench_ring(dist, rank, world, steps=3, warmup=1, B=1, H=16, n_local=8192, d=128):
    """Times the ring forward; returns the dict bench.py embeds as "ring" (None when world == 1
    and the library is absent).  B=1, H=16 is this build's stated choice for the BASELINE ring
    config, which names only N=65536, d=128."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    mk = lambda: (torch.rand(B, H, n_local, d, device=dev, generator=g) - 0.5).to(torch.bfloat16)
    Q, K, V = mk(), mk(), mk()
    scale = 1.0 / math.sqrt(d)
    N = n_local * world
    flops = 4.0 * B * H * N * N * d

    def barrier():
        if world > 1:
            dist.barrier()

    def time_it(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps

    out = {"config": {"workload": f"ring FA2 forward bf16, seq-sharded N={N} (N/P={n_local}), d={d}, B={B}, H={H}",
                      "n_gpus": world}, "unit": "TFLOP/s", "transport": "rccl-native (libfa2_ring_mi355x.so)"}
    # ONE code path: the native library (RCCL called from C++).  A failure is reported as such -- never replaced by
    # a number from another transport.  Both schedules are listed; `value` is chosen by a FIXED rule, not by speed:
    # the mesh schedule (owner-direct fetch over the 7 xGMI links, this build's default for P > 2), else the relay.
    results = {}
    ctx = RingContext(dist, rank, world)
    try:
        for name in ("relay", "mesh") if world > 1 else ("relay",):
            O = torch.empty_like(Q)
            L = torch.empty(B, H, n_local, dtype=torch.float32, device=dev)
            sec = time_it(lambda: ring_attention_forward(ctx, Q, K, V, scale, schedule=name, O_local=O, L_local=L))
            results[name] = {"ms": round(sec * 1e3, 4), "tflops": round(flops / sec / 1e12, 2)}
        # the ring BACKWARD beside it (past the reference, whose ring is forward-only): K/V fetched from their owners, the
        # block kernels of fa2_backward_block per resident shard (dense square blocks of this length run the single
        # five-product kernel), dK/dV pieces exchanged while the next block runs.  10 B H N^2 d flops.
        try:
            dO = (torch.rand(B, H, n_local, d, device=dev, generator=g) - 0.5).mul_(0.4).to(torch.bfloat16)
            sec = time_it(lambda: ring_attention_backward(ctx, Q, K, V, O, L, dO, scale))
            out["backward"] = {"ms": round(sec * 1e3, 4), "tflops": round(2.5 * flops / sec / 1e12, 2),
                               "flops": "10 B H N^2 d", "pct_mfma_peak": round(100.0 * 2.5 * flops / sec / 1e12 / world / 2516.6, 2)}
        except Exception as e:          # a side figure of a side figure: never takes the forward's numbers down
            out["backward"] = {"error": repr(e)}
    finally:
        ctx.close()
    out["schedules"] = results
    best = "mesh" if world > 2 else "relay"
    out["value"] = results[best]["tflops"]
    out["schedule"] = best
    out["pct_mfma_peak"] = round(100.0 * results[best]["tflops"] / world / 2516.6, 2)
    return out
