"""GPU (one device): the native ring library end to end with a 1-rank RCCL communicator --
library loading next to torch's RCCL, context creation from a unique id, the self
send/receive exchange primitive, and ring forward == plain forward at P = 1.  P > 1 needs
several GPUs and is covered by tests/test_ring_gloo.py (schedule) + the driver's scaling run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_ring_single_rank_native():
    import cuda_flashattention_amd as fa
    from cuda_flashattention_amd import ring
    ctx = ring.RingContext(None, 0, 1)
    try:
        B, H, n, d = 1, 4, 512, 128
        g = torch.Generator().manual_seed(1)
        Q, K, V = ((torch.rand(B, H, n, d, generator=g) - 0.5).bfloat16().cuda() for _ in range(3))
        for sched in ("relay", "mesh"):
            O, L = ring.ring_attention_forward(ctx, Q, K, V, schedule=sched)
            O2, L2 = fa.flash_attention_2_forward(Q, K, V)
            torch.cuda.synchronize()
            assert torch.equal(O, O2) and torch.equal(L, L2)
        # exchange primitive: with one rank, next == prev == self
        a = torch.arange(4096, dtype=torch.float32, device="cuda")
        b = a * 2
        ra, rb = torch.zeros_like(a), torch.zeros_like(b)
        st = ring.ring_lib().fa2_ring_exchange_kv(ctx._h, a.data_ptr(), ra.data_ptr(), b.data_ptr(),
                                                  rb.data_ptr(), a.numel() * 4,
                                                  torch.cuda.current_stream().cuda_stream)
        assert st == 0
        torch.cuda.synchronize()
        assert torch.equal(ra, a) and torch.equal(rb, b)
    finally:
        ctx.close()


def test_ring_reference_signature_fp32_single_rank(golden):
    """ring_attention_forward(Q_local, K_local, V_local, O_local, L_local, total, local, d, scale,
    comm, rank, nranks) with nranks = 1 on the reference's ring data (N=5096, d=64, scale 1)."""
    import ctypes
    from cuda_flashattention_amd import ring
    from oracle import recipes
    lib = ring.ring_lib()
    rccl = ctypes.CDLL("librccl.so.1")
    ident = ctypes.create_string_buffer(128)
    assert lib.fa2_ring_get_unique_id(ident) == 0
    comm = ctypes.c_void_p()

    class UID(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UID.from_buffer_copy(ident.raw)
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        g = golden("k5_ring_pattern_rows.npz")
        N, d = int(g["N"]), int(g["d"])
        Q, K, V = (torch.from_numpy(x).cuda() for x in recipes.ring_pattern(N, d))
        K0 = K.clone()
        O = torch.empty_like(Q)
        L = torch.empty(N, device="cuda")
        st = lib.ring_attention_forward(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
                                        N, N, d, 1.0, comm, 0, 1)
        assert st == 0
        assert torch.equal(K, K0)
        assert recipes.compare_outputs(recipes.ring_pattern_expected(N, d), O.cpu().numpy(), rtol=5e-3, atol=1.0) == 0
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("P,B,H,N,d", [(2, 1, 2, 512, 128), (4, 1, 3, 1024, 64), (8, 2, 8, 2048, 128)])
def test_causal_zigzag_blocks_on_one_gpu(P, B, H, N, d):
    """The causal ring's per-rank work -- the strided / causal step kernel and the finalize pass -- for P
    virtual ranks played one after the other on this GPU (the K/V 'exchange' is a pointer), against the
    single-GPU causal forward on the same inputs.  (The RCCL transport itself needs P GPUs.)"""
    import cuda_flashattention_amd as fa
    from cuda_flashattention_amd import ring
    g = torch.Generator().manual_seed(5)
    mk = lambda: (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16().cuda()
    Q, K, V = mk(), mk(), mk()
    s = 1.0 / d ** 0.5
    Oref, Lref = fa.flash_attention_2_forward(Q, K, V, s, causal=True)
    O = torch.empty_like(Q)
    L = torch.empty(B, H, N, device="cuda")
    shards = []
    for r in range(P):
        rows = torch.tensor(ring.zigzag_rows(N, r, P), device="cuda")
        shards.append((rows, Q[:, :, rows].contiguous(), K[:, :, rows].contiguous(), V[:, :, rows].contiguous()))
    for r in range(P):
        rows, Ql, _, _ = shards[r]
        Ol = torch.empty_like(Ql)
        Ll = torch.empty(B, H, Ql.shape[2], device="cuda")
        Ml = torch.empty_like(Ll)
        Oacc = torch.empty(Ql.shape, dtype=torch.float32, device="cuda")
        for step in range(P):
            owner = ring.kv_owner(r, step, P)
            ring._gpu_block(Ql, shards[owner][2], shards[owner][3], Ol, Ll, Oacc, Ml, s, ring.causal_block_kind(r, owner))
        ring._gpu_finalize(Ol, Ll, Oacc, Ml)
        O[:, :, rows] = Ol
        L[:, :, rows] = Ll
    torch.cuda.synchronize()
    a, b = O.float().cpu().numpy(), Oref.float().cpu().numpy()
    # two bf16 evaluations with different block orders: each carries the bf16 rounding of P (2e-3 against the
    # oracle), so they differ by about sqrt(2) of that; L comes from fp32 sums and must agree closely
    assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 5e-3
    assert np.abs(L.cpu().numpy() - Lref.cpu().numpy()).max() <= 1e-4


def test_causal_ring_one_rank_is_local_causal():
    import cuda_flashattention_amd as fa
    from cuda_flashattention_amd import ring
    B, H, N, d = 1, 4, 768, 128
    g = torch.Generator().manual_seed(6)
    mk = lambda: (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16().cuda()
    Q, K, V = mk(), mk(), mk()
    ctx = ring.RingContext(None, 0, 1)
    try:
        O, L = ring.ring_attention_forward(ctx, Q, K, V, causal=True)
        Oref, Lref = fa.flash_attention_2_forward(Q, K, V, causal=True)
        torch.cuda.synchronize()
        assert np.linalg.norm((O.float() - Oref.float()).cpu().numpy()) / np.linalg.norm(Oref.float().cpu().numpy()) <= 1e-3
        assert (L - Lref).abs().max().item() <= 1e-4
        with pytest.raises(fa._capi.FA2Error):          # fp32 has no causal ring
            ring.ring_attention_forward(ctx, Q.float(), K.float(), V.float(), causal=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("P,B,H,N,d", [(2, 1, 2, 512, 128), (4, 1, 3, 1024, 64)])
def test_ring_backward_blocks_on_one_gpu(P, B, H, N, d):
    """The ring backward's per-rank work for P virtual ranks played one after the other on this GPU: the ordinary
    backward kernels on (local rows) x (one shard of keys) with the WHOLE sequence's L, dQ summed per rank and
    dK/dV per owner in fp32 -- against the single-GPU backward."""
    import cuda_flashattention_amd as fa
    g = torch.Generator().manual_seed(7)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16().cuda()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    s = 1.0 / d ** 0.5
    O, L = fa.flash_attention_2_forward(Q, K, V, s)
    ref = fa.flash_attention_2_backward(Q, K, V, O, L, dO, s)
    n = N // P
    sh = lambda t, r: t[:, :, r * n:(r + 1) * n].contiguous()
    aq = [None] * P
    ak = [torch.zeros(B, H, n, d, device="cuda") for _ in range(P)]
    av = [torch.zeros(B, H, n, d, device="cuda") for _ in range(P)]
    for r in range(P):
        for o in range(P):
            dq, dk, dv = fa.flash_attention_2_backward(sh(Q, r), sh(K, o), sh(V, o), sh(O, r), L[:, :, r * n:(r + 1) * n].contiguous(),
                                                       sh(dO, r), s)
            aq[r] = dq.float() if aq[r] is None else aq[r] + dq.float()
            ak[o] += dk.float()
            av[o] += dv.float()
    torch.cuda.synchronize()
    got = [torch.cat(x, dim=2) for x in (aq, ak, av)]
    for a, b in zip(got, ref):
        a, b = a.cpu().numpy(), b.float().cpu().numpy()
        assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 5e-3


def test_ring_backward_one_rank_native():
    import cuda_flashattention_amd as fa
    from cuda_flashattention_amd import ring
    B, H, N, d = 1, 4, 512, 128
    g = torch.Generator().manual_seed(8)
    mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16().cuda()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    ctx = ring.RingContext(None, 0, 1)
    try:
        O, L = ring.ring_attention_forward(ctx, Q, K, V)
        dQ, dK, dV = ring.ring_attention_backward(ctx, Q, K, V, O, L, dO)
        ref = fa.flash_attention_2_backward(Q, K, V, O, L, dO)
        torch.cuda.synchronize()
        for a, b in zip((dQ, dK, dV), ref):
            assert torch.equal(a, b)              # one rank: bf16 -> fp32 -> bf16 of the same kernels' outputs
    finally:
        ctx.close()
