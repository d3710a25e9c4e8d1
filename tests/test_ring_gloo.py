"""CPU, world_size 2 and 3 over gloo: the ring schedule (who computes on whose shard at which
step, the double-buffered exchange, the resumable softmax state) end to end, with the oracle's
ring step (ring_attention_kernel.cu:67-137 restated) standing in for the HIP step kernel.
Compares against one-shot attention on the gathered sequence, like 04_ring_attention.cu does
with MPI_Gather + compare_outputs."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_step(Q, K, V, O, L, Oacc, M, scale, first, last):
    """step_fn for ring_attention_forward_p2p on CPU fp32 tensors [N/P, d]: the state lives in
    (O, L, M) exactly as in the reference (O un-normalised until the last step)."""
    import oracle
    if first:
        O.zero_()
        L.zero_()
        M.fill_(float("-inf"))
    oracle.ring_step(Q.numpy(), K.numpy(), V.numpy(), O.numpy(), L.numpy(), M.numpy(), float(scale), last)


def _worker(rank, world, port, N, d, scale, pattern, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuda_flashattention_amd import ring
        from oracle import recipes
        if pattern:
            Q, K, V = recipes.ring_pattern(N, d)
        else:
            rng = np.random.default_rng(0)
            Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
        lo, hi = ring.shard_rows(N, rank, world)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi]))
        Kl, Vl = t(K), t(V)
        K0, V0 = Kl.clone(), Vl.clone()
        O, L = ring.ring_attention_forward_p2p(dist, t(Q), Kl, Vl, scale, step_fn=_oracle_step)
        assert torch.equal(Kl, K0) and torch.equal(Vl, V0)          # caller's shards preserved
        gathered = [torch.empty_like(O) for _ in range(world)]
        dist.all_gather(gathered, O)                                  # MPI_Gather, rank order = row order
        gl = [torch.empty_like(L) for _ in range(world)]
        dist.all_gather(gl, L)
        if rank == 0:
            ret["O"] = torch.cat(gathered).numpy()
            ret["L"] = torch.cat(gl).numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,d", [(2, 128, 32), (3, 96, 16), (2, 130, 64)])
def test_ring_schedule_matches_one_shot(world, N, d):
    import oracle
    scale = 1.0 / np.sqrt(d)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), N, d, scale, False, ret), nprocs=world, join=True)
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
        mp.spawn(_worker, args=(world, _free_port(), N, d, 1.0, True, ret), nprocs=world, join=True)
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


# ----------------------------------------------------------------------------- causal zig-zag ring
def _oracle_causal_block(Q, K, V, O, L, Oacc, M, scale, kind):
    """block_fn for ring_attention_forward_causal_p2p on CPU fp32 tensors [2c, d].  State in (O, L, M) as in
    _oracle_step.  "local": exact causal attention over the local rows, stored as the equivalent state
    (acc = O, l = 1, m = LSE); the other two blocks are plain ring steps on row slices."""
    import oracle
    n = Q.shape[0]
    c = n // 2
    if kind == "local":
        o, lse = oracle.attention_forward(Q.numpy(), K.numpy(), V.numpy(), float(scale), causal=True)
        O.copy_(torch.from_numpy(np.asarray(o, dtype=np.float32)))
        M.copy_(torch.from_numpy(np.asarray(lse, dtype=np.float32)))
        L.fill_(1.0)
    elif kind == "first_keys":
        oracle.ring_step(Q.numpy(), K[:c].numpy(), V[:c].numpy(), O.numpy(), L.numpy(), M.numpy(), float(scale), False)
    else:
        oracle.ring_step(Q[c:].numpy(), K.numpy(), V.numpy(), O[c:].numpy(), L[c:].numpy(), M[c:].numpy(), float(scale), False)


def _oracle_finalize(O, L, Oacc, M):
    O /= L[:, None]
    L.copy_(M + torch.log(L))


def _causal_worker(rank, world, port, N, d, scale, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuda_flashattention_amd import ring
        rng = np.random.default_rng(1)
        Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
        rows = ring.zigzag_rows(N, rank, world)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[rows]))
        O, L = ring.ring_attention_forward_causal_p2p(dist, t(Q), t(K), t(V), scale, block_fn=_oracle_causal_block,
                                                      finalize_fn=_oracle_finalize)
        go = [torch.empty_like(O) for _ in range(world)]
        gl = [torch.empty_like(L) for _ in range(world)]
        dist.all_gather(go, O)
        dist.all_gather(gl, L)
        if rank == 0:
            Of = np.empty((N, d), np.float32)
            Lf = np.empty(N, np.float32)
            for r in range(world):                      # undo the zig-zag order
                rr = ring.zigzag_rows(N, r, world)
                Of[rr] = go[r].numpy()
                Lf[rr] = gl[r].numpy()
            ret["O"], ret["L"] = Of, Lf
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,d", [(2, 128, 32), (3, 96, 16)])
def test_causal_zigzag_ring_matches_one_shot(world, N, d):
    """Causal ring with zig-zag sharding (past the reference): every rank's blocks --
    local causal, first-chunk keys for all rows, all keys for the second-chunk rows -- add up to causal
    attention over the whole sequence.  No counterpart in the reference: pinned by the oracle's own
    masked-dense causal forward."""
    import oracle
    scale = 1.0 / np.sqrt(d)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_causal_worker, args=(world, _free_port(), N, d, scale, ret), nprocs=world, join=True)
        O, L = ret["O"], ret["L"]
    rng = np.random.default_rng(1)
    Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
    Or, Lr = oracle.attention_forward(Q, K, V, float(scale), causal=True)
    assert np.abs(O - Or).max() < 5e-6
    assert np.abs(L - Lr).max() < 5e-6


def test_zigzag_helpers():
    from cuda_flashattention_amd import ring
    for P in (1, 2, 4, 8):
        N = 16 * P
        seen = sorted(i for r in range(P) for i in ring.zigzag_rows(N, r, P))
        assert seen == list(range(N))                   # a partition of the sequence
        for r in range(P):
            kinds = [ring.causal_block_kind(r, ring.kv_owner(r, s, P)) for s in range(P)]
            assert kinds[0] == "local" and kinds.count("local") == 1
            assert kinds.count("first_keys") == r and kinds.count("second_rows") == P - 1 - r
    with pytest.raises(ValueError):
        ring.zigzag_rows(100, 0, 3)


# ----------------------------------------------------------------------------- ring backward
def _numpy_bwd_block(Q, K, V, O, L, dO, scale):
    """block_fn for ring_attention_backward_p2p on CPU fp32 tensors [n, d]: the local rows' gradient pieces against
    one shard of keys, given the log-sum-exp L of the WHOLE sequence (flash_attention_backward_kernel.cu:47-246
    restricted to a key range): P = exp(S - L), dP = dO V^T, dS = P (dP - D), D = rowsum(dO O)."""
    q, k, v, o, l, g = (t.numpy().astype(np.float64) for t in (Q, K, V, O, L, dO))
    p = np.exp(scale * q @ k.T - l[:, None])
    ds = p * (g @ v.T - (g * o).sum(1)[:, None])
    f = lambda a: torch.from_numpy(a.astype(np.float32))
    return f(scale * ds @ k), f(scale * ds.T @ q), f(p.T @ g)


def _bwd_worker(rank, world, port, N, d, scale, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from cuda_flashattention_amd import ring
        rng = np.random.default_rng(2)
        Q, K, V, dO = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(4))
        O, L = oracle.attention_forward(Q, K, V, float(scale))          # the ring forward's outputs, whole sequence
        lo, hi = ring.shard_rows(N, rank, world)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)[lo:hi]))
        dQ, dK, dV = ring.ring_attention_backward_p2p(dist, t(Q), t(K), t(V), t(O), t(L), t(dO), scale,
                                                      block_fn=_numpy_bwd_block)
        for name, x in (("dQ", dQ), ("dK", dK), ("dV", dV)):
            parts = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(parts, x)
            if rank == 0:
                ret[name] = torch.cat(parts).numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,d", [(2, 128, 32), (3, 96, 16)])
def test_ring_backward_matches_one_shot(world, N, d):
    """Ring backward (past the reference): per-shard gradient pieces, dQ summed locally and dK/dV summed at the
    shard's owner, equal the one-shot backward over the whole sequence (the oracle's restatement of
    naive_attention_backward, util/naive_attention.h:84-161)."""
    import oracle
    scale = 1.0 / np.sqrt(d)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bwd_worker, args=(world, _free_port(), N, d, scale, ret), nprocs=world, join=True)
        got = {k: ret[k] for k in ("dQ", "dK", "dV")}
    rng = np.random.default_rng(2)
    Q, K, V, dO = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(4))
    dQ, dK, dV = oracle.attention_backward(Q, K, V, dO, float(scale))
    for name, ref in (("dQ", dQ), ("dK", dK), ("dV", dV)):
        assert np.abs(got[name] - np.asarray(ref)).max() < 5e-6, name
