"""CPU, world_size 2 and 3 over gloo: the C++ ring schedules of libfa2_ring_mi355x.so (csrc/ring/fa2_ring.cpp) across
REAL processes.  Each process installs a backend (tests/ring_sim.py: EagerWorld) whose transport is
torch.distributed point-to-point over gloo and whose per-step arithmetic is the oracle's ring step
(ring_attention_kernel.cu:67-137 restated) on host memory -- so what is exercised is the shipped schedule (who
computes on whose shard at which step, slots, peers, the resumable softmax state), not a Python copy of it.
Results are gathered and compared against one-shot attention on the whole sequence, like 04_ring_attention.cu does
with MPI_Gather + compare_outputs."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _gather(x, world):
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x)
    return [p.numpy() for p in parts]


def _fwd_worker(rank, world, port, N, d, scale, pattern, schedule, ret):
    _init(rank, world, port)
    try:
        import ring_sim as rs
        from oracle import recipes
        if pattern:
            Q, K, V = recipes.ring_pattern(N, d)
        else:
            rng = np.random.default_rng(0)
            Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)
        q, k, v = (np.ascontiguousarray(a[sl], dtype=np.float32) for a in (Q, K, V))
        k0, v0 = k.copy(), v.copy()
        o = np.full((n, d), np.nan, np.float32)
        l = np.full(n, np.nan, np.float32)
        w = rs.EagerWorld(dist, rank, world)
        lib = w.lib
        need = lib.fa2_ring_workspace_bytes(1, 1, n, d, rs.FA2_DTYPE_F32, world, schedule)
        ws = np.full(max(need, 256), 0xFF, np.uint8)
        st = lib.fa2_ring_attention_forward(w.ctx(), q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data, l.ctypes.data,
                                            1, 1, N, n, d, scale, rs.FA2_DTYPE_F32, schedule, ws.ctypes.data, need, 1)
        assert st == 0 and not w.errors, (st, w.errors)
        assert np.array_equal(k, k0) and np.array_equal(v, v0)            # caller's shards preserved
        go, gl = _gather(torch.from_numpy(o), world), _gather(torch.from_numpy(l), world)
        if rank == 0:                                                     # MPI_Gather, rank order = row order
            ret["O"], ret["L"] = np.concatenate(go), np.concatenate(gl)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,d,schedule", [(2, 128, 32, 0), (3, 96, 16, 0), (2, 130, 64, 1), (3, 96, 16, 1)])
def test_ring_schedule_matches_one_shot(world, N, d, schedule):
    import oracle
    scale = 1.0 / np.sqrt(d)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_fwd_worker, args=(world, _free_port(), N, d, scale, False, schedule, ret), nprocs=world, join=True)
        O, L = ret["O"], ret["L"]
    rng = np.random.default_rng(0)
    Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
    Or, Lr = oracle.naive_forward_pass(Q, K, V, float(scale))
    assert np.abs(O - Or).max() < 2e-6
    assert np.abs(L - Lr).max() < 2e-6


def test_ring_reference_pattern_two_ranks():
    """The reference's own ring test data (create_simple_test_data, scale 1) at a reduced length
    (N=512 instead of 5096 to stay in CPU seconds), P=2 as run.sh:2, judged with its criterion."""
    from oracle import recipes
    N, d, world = 512, 64, 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_fwd_worker, args=(world, _free_port(), N, d, 1.0, True, 0, ret), nprocs=world, join=True)
        O = ret["O"]
    assert recipes.compare_outputs(recipes.ring_pattern_expected(N, d), O, rtol=5e-3, atol=1.0) == 0


def test_shard_helpers():
    from cuda_flashattention_amd import ring
    assert ring.shard_rows(5096, 1, 2) == (2548, 5096)
    with pytest.raises(ValueError):
        ring.shard_rows(5096, 0, 3)                     # "seq_len must be divisible by nranks!"
    # every rank sees every shard exactly once, starting with its own (ring_attention_kernel.cu:198)
    for P in (2, 4, 8):
        for r in range(P):
            owners = [ring.kv_owner(r, s, P) for s in range(P)]
            assert owners[0] == r and sorted(owners) == list(range(P))
    for P in (1, 2, 4, 8):
        N = 16 * P
        assert sorted(i for r in range(P) for i in ring.zigzag_rows(N, r, P)) == list(range(N))     # a partition
    with pytest.raises(ValueError):
        ring.zigzag_rows(100, 0, 3)


# ----------------------------------------------------------------------------- causal forward + backward, bf16 state layout
def _bf16_inputs(N, d, seed):
    import ring_sim as rs
    rng = np.random.default_rng(seed)
    Q, K, V = (rs.round_bf16(rng.uniform(-0.5, 0.5, (1, 1, N, d)).astype(np.float32)).reshape(1, 1, N, d) for _ in range(3))
    dO = rs.round_bf16(rng.uniform(-0.2, 0.2, (1, 1, N, d)).astype(np.float32)).reshape(1, 1, N, d)
    return Q, K, V, dO


def _train_worker(rank, world, port, N, d, scale, causal, ret):
    """Ring forward, then ring backward fed by the forward's own outputs, as a training step does."""
    _init(rank, world, port)
    try:
        import ring_sim as rs
        Q, K, V, dO = _bf16_inputs(N, d, 3)
        n = N // world
        rows = rs.zigzag_rows(N, rank, world) if causal else list(range(rank * n, (rank + 1) * n))
        st16 = lambda a: rs.f32_to_bf16(np.ascontiguousarray(a[:, :, rows])).reshape(1, 1, n, d)
        q, k, v, g = st16(Q), st16(K), st16(V), st16(dO)
        o = np.zeros((1, 1, n, d), np.uint16)
        l = np.zeros((1, 1, n), np.float32)
        w = rs.EagerWorld(dist, rank, world)
        lib, ctx = w.lib, w.ctx()
        need = lib.fa2_ring_workspace_bytes(1, 1, n, d, rs.FA2_DTYPE_BF16, world, rs.RELAY)
        ws = np.full(max(need, 256), 0xFF, np.uint8)
        fwd = lib.fa2_ring_attention_forward_causal if causal else lib.fa2_ring_attention_forward
        st = fwd(ctx, q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data, l.ctypes.data, 1, 1, N, n, d, scale,
                 rs.FA2_DTYPE_BF16, rs.RELAY, ws.ctypes.data, need, 1)
        assert st == 0 and not w.errors, (st, w.errors)
        outs = [np.zeros((1, 1, n, d), np.uint16) for _ in range(3)]
        need = lib.fa2_ring_backward_workspace_bytes(1, 1, n, d, rs.FA2_DTYPE_BF16, world)
        ws = np.full(max(need, 256), 0xFF, np.uint8)
        bwd = lib.fa2_ring_attention_backward_causal if causal else lib.fa2_ring_attention_backward
        st = bwd(ctx, q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data, l.ctypes.data, g.ctypes.data,
                 outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, 1, 1, N, n, d, scale, rs.FA2_DTYPE_BF16,
                 ws.ctypes.data, need, 1)
        assert st == 0 and not w.errors, (st, w.errors)
        f32 = lambda a: torch.from_numpy(rs.bf16_to_f32(a).reshape(n, d).copy())
        parts = {name: _gather(f32(x), world) for name, x in (("O", o), ("dQ", outs[0]), ("dK", outs[1]), ("dV", outs[2]))}
        parts["L"] = _gather(torch.from_numpy(l.reshape(n).copy()), world)
        if rank == 0:
            for name, ps in parts.items():
                full = np.empty((N,) + ps[0].shape[1:], np.float32)
                for r in range(world):
                    rr = rs.zigzag_rows(N, r, world) if causal else list(range(r * n, (r + 1) * n))
                    full[rr] = ps[r]
                ret[name] = full
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,d,causal", [(2, 128, 32, False), (3, 96, 16, False), (2, 128, 32, True), (3, 96, 16, True)])
def test_ring_forward_backward_matches_one_shot(world, N, d, causal):
    """Forward then backward over gloo, plain and causal (zig-zag sharding; both past the reference, SURVEY 8f rank 2:
    pinned by the oracle's own masked-dense forms).  bf16 storage, fp32 state and sums: bf16 gates of DESIGN.md."""
    import oracle
    scale = 1.0 / np.sqrt(d)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_train_worker, args=(world, _free_port(), N, d, scale, causal, ret), nprocs=world, join=True)
        got = {k: ret[k] for k in ("O", "L", "dQ", "dK", "dV")}
    Q, K, V, dO = _bf16_inputs(N, d, 3)
    Or, Lr = oracle.attention_forward(Q, K, V, float(scale), causal=causal)
    ref = oracle.attention_backward(Q, K, V, dO, float(scale), causal=causal)
    rel = lambda a, b: np.linalg.norm(a - b.reshape(a.shape)) / np.linalg.norm(b)
    assert rel(got["O"], Or) < 3e-3
    assert np.abs(got["L"] - Lr.reshape(-1)).max() < 1e-5
    for name, r in zip(("dQ", "dK", "dV"), ref):
        assert rel(got[name], r) < 5e-3, name
