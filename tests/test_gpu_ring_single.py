"""GPU (one device): the native ring library end to end with a 1-rank RCCL communicator --
library loading next to torch's RCCL, context creation from a unique id, the self
send/receive exchange primitive, and ring forward == plain forward at P = 1 (and == the oracle).  P > 1: the same
C++ schedules run at P = 2, 4, 8 on this one GPU in tests/test_gpu_ring_loopback.py (loopback transport), on the CPU
in tests/test_ring_sim.py and across processes in tests/test_ring_gloo.py."""
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
        import oracle
        f = lambda t: t.float().cpu().numpy()
        Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), float(1.0 / d ** 0.5))
        assert np.linalg.norm(f(O) - Or) / np.linalg.norm(Or) <= 5e-3 and np.abs(L.cpu().numpy() - Lr).max() <= 1e-4
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
        import oracle
        torch.cuda.synchronize()
        f = lambda t: t.float().cpu().numpy()
        Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), float(1.0 / d ** 0.5), causal=True)
        assert np.linalg.norm(f(O) - Or) / np.linalg.norm(Or) <= 5e-3
        assert np.abs(L.cpu().numpy() - Lr).max() <= 1e-4
        with pytest.raises(fa._capi.FA2Error):          # fp32 has no causal ring
            ring.ring_attention_forward(ctx, Q.float(), K.float(), V.float(), causal=True)
    finally:
        ctx.close()


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
        # the ring's dense square blocks go through fa2_backward_block, which -- like fa2_backward -- gives an eligible
        # block (d = 128, local length % 256 == 0) to the single five-product kernel: fa2_backward is the bit-exact reference
        ref = [torch.empty_like(Q) for _ in range(3)]
        ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
        fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=ref[0], dK=ref[1], dV=ref[2], workspace=ws)
        torch.cuda.synchronize()
        for a, b in zip((dQ, dK, dV), ref):
            assert torch.equal(a, b)              # one rank: bf16 -> fp32 -> bf16 of the same kernels' outputs
    finally:
        ctx.close()
