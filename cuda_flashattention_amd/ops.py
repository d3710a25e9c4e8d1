"""Host-side mirror of the reference's operator interface for the FA2 hot path.

The reference's interface is three free functions taking raw device pointers
(flash_attention_2_forward / flash_attention_2_backward / ring_attention_forward, see
include/fa2_mi355x.h for file:line).  These wrappers keep the same names and argument
meaning, take torch tensors only as *device memory* (pointer + shape), and call the C ABI
through ctypes.  torch is plumbing here: allocation, streams, nothing numerical.
"""
import math

import torch

from . import _capi
from ._capi import FA2_DTYPE_BF16, FA2_DTYPE_F32, FA2_DTYPE_FP8_E4M3, check


def _dtype_code(t):
    if t.dtype == torch.bfloat16:
        return FA2_DTYPE_BF16
    if t.dtype == torch.float32:
        return FA2_DTYPE_F32
    if t.dtype == torch.float8_e4m3fn:
        return FA2_DTYPE_FP8_E4M3
    raise TypeError(f"unsupported dtype {t.dtype}: bf16, fp32 or float8_e4m3fn (forward only) expected")


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


def _bhnd(x, name):
    if not isinstance(x, torch.Tensor):
        raise ValueError(f"{name} must be a torch tensor")
    if not x.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if not x.is_contiguous():
        raise ValueError(f"{name} must be contiguous [B][H][N][d] (got strides {tuple(x.stride())}; call .contiguous())")
    if x.dim() == 2:
        return 1, 1, x.shape[0], x.shape[1]
    if x.dim() == 4:
        return tuple(x.shape)
    raise ValueError(f"{name}: expected [N,d] or [B,H,N,d], got {tuple(x.shape)}")


def _like(x, name, ref, shape, dtype):
    """The C ABI takes raw pointers: everything it will read or write as a dense [B][H][N][d] (or [B][H][N]) tensor is
    checked here -- device, contiguity, shape, dtype -- instead of becoming a silent out-of-bounds access."""
    if _bhnd(x, name) != shape and tuple(x.shape) != shape:
        raise ValueError(f"{name}: shape {tuple(x.shape)} does not match {shape}")
    if x.dtype != dtype:
        raise ValueError(f"{name}: dtype {x.dtype}, expected {dtype}")
    if x.device != ref.device:
        raise ValueError(f"{name} is on {x.device}, expected {ref.device}")
    return x


def _rows(x, name, ref, B, H, N):
    if not isinstance(x, torch.Tensor) or not x.is_cuda or not x.is_contiguous():
        raise ValueError(f"{name} must be a contiguous device tensor")
    if x.dtype != torch.float32 or x.numel() != B * H * N or x.device != ref.device:
        raise ValueError(f"{name}: expected fp32 [B,H,N] = {B * H * N} elements on {ref.device}, got {x.dtype} {tuple(x.shape)}")
    return x


def forward_fp8_workspace(B, H, N, d, device="cuda"):
    """Scratch for the fp8 forward (V transposed + key-norm maxima): allocate once, pass as workspace=."""
    return torch.empty(_capi.lib().fa2_forward_fp8_workspace_bytes(B, H, N, d), dtype=torch.uint8, device=device)


def flash_attention_2_forward(Q, K, V, softmax_scale=None, causal=False, O=None, L=None, stream=None, workspace=None,
                              descale=None):
    """O, L = FA2 forward.  Mirrors flash_attention_2_forward(Q,K,V,O,L,seq_len,head_dim,scale)
    (reference 02_forward/flash_attention_kernel.cu:300-309) with B,H,dtype,causal,stream added.
    Tensors [N,d] or [B,H,N,d]; L is fp32 [.., N] natural-log LSE.  fp8 (e4m3) inputs only: workspace= a caller-owned
    scratch tensor (forward_fp8_workspace; without it every call takes one from the stream-ordered allocator), descale=
    (q, k, v) per-tensor descales of tensors stored as x / descale (fa2_forward_fp8_scaled)."""
    B, H, N, d = shape = _bhnd(Q, "Q")
    for n, t in (("K", K), ("V", V)):
        _like(t, n, Q, shape, Q.dtype)
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    odt = torch.bfloat16 if Q.dtype == torch.float8_e4m3fn else Q.dtype      # fp8 inputs (OCP e4m3, d = 128) produce a bf16 O
    if O is None:
        O = torch.empty(Q.shape, dtype=odt, device=Q.device)
    if L is None:
        L = torch.empty(Q.shape[:-1], dtype=torch.float32, device=Q.device)
    _like(O, "O", Q, shape, odt)
    _rows(L, "L", Q, B, H, N)
    if Q.dtype == torch.float8_e4m3fn and (workspace is not None or descale is not None):
        if workspace is None:
            workspace = forward_fp8_workspace(B, H, N, d, Q.device)
        if not workspace.is_cuda or not workspace.is_contiguous() or workspace.device != Q.device:
            raise ValueError("workspace must be a contiguous device tensor on Q's device")
        dq, dk, dv = (1.0, 1.0, 1.0) if descale is None else (float(x) for x in descale)
        st = _capi.lib().fa2_forward_fp8_scaled(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(), B, H, N, d, scale,
                                                dq, dk, dv, 1 if causal else 0, workspace.data_ptr(),
                                                workspace.numel() * workspace.element_size(), _stream_ptr(stream))
        check(st, "fa2_forward_fp8_scaled")
        return O, L
    if workspace is not None or descale is not None:
        raise ValueError("workspace= / descale= belong to the fp8 (float8_e4m3fn) forward")
    st = _capi.lib().fa2_forward(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
                                 B, H, N, d, scale, _dtype_code(Q), 1 if causal else 0, _stream_ptr(stream))
    check(st, "fa2_forward")
    return O, L


def flash_attention_2_backward(Q, K, V, O, L, dO, softmax_scale=None, causal=False,
                               dQ=None, dK=None, dV=None, workspace=None, stream=None, phases=7):
    """dQ, dK, dV = FA2 backward.  Mirrors flash_attention_2_backward(Q,K,V,O,L,dO,dQ,dK,dV,...)
    (reference 02_backward/flash_attention_backward_kernel.cu:249-262).  dO must be contiguous (autograd often hands
    over an expanded or transposed view: call .contiguous() on it first -- the C ABI reads a dense tensor)."""
    B, H, N, d = shape = _bhnd(Q, "Q")
    gdt = torch.bfloat16 if Q.dtype == torch.float8_e4m3fn else Q.dtype       # what the forward produced for fp8 inputs
    for n, t, dt in (("K", K, Q.dtype), ("V", V, Q.dtype), ("O", O, gdt), ("dO", dO, gdt)):
        _like(t, n, Q, shape, dt)
    _rows(L, "L", Q, B, H, N)
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    dQ = torch.empty_like(Q) if dQ is None else _like(dQ, "dQ", Q, shape, Q.dtype)
    dK = torch.empty_like(K) if dK is None else _like(dK, "dK", Q, shape, Q.dtype)
    dV = torch.empty_like(V) if dV is None else _like(dV, "dV", Q, shape, Q.dtype)
    lib = _capi.lib()
    need = lib.fa2_backward_workspace_bytes(B, H, N, d, _dtype_code(Q))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=Q.device)
    if not isinstance(workspace, torch.Tensor) or not workspace.is_cuda or not workspace.is_contiguous() or workspace.device != Q.device:
        raise ValueError("workspace must be a contiguous device tensor on Q's device")
    st = lib.fa2_backward_phases(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
                                 dO.data_ptr(), dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                                 B, H, N, d, scale, _dtype_code(Q), 1 if causal else 0,
                                 workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                 _stream_ptr(stream), int(phases))
    check(st, "fa2_backward")
    return dQ, dK, dV


def forward_step(Q, K, V, O, L, Oacc, M, softmax_scale, first, last, stream=None):
    """One resumable ring step (ring_attention_forward_kernel, ring_attention_kernel.cu:13-140)."""
    B, H, Nq, d = qshape = _bhnd(Q, "Q")
    _, _, Nk, _ = kshape = _bhnd(K, "K")
    if kshape != (B, H, Nk, d) or K.dtype != Q.dtype:
        raise ValueError("K must match Q in B, H, d and dtype")
    _like(V, "V", Q, kshape, Q.dtype)
    if O is not None:
        _like(O, "O", Q, qshape, Q.dtype)
    _rows(L, "L", Q, B, H, Nq)
    if Oacc is not None:
        _like(Oacc, "Oacc", Q, qshape, torch.float32)
    if M is not None:
        _rows(M, "M", Q, B, H, Nq)
    st = _capi.lib().fa2_forward_step(Q.data_ptr(), K.data_ptr(), V.data_ptr(),
                                      O.data_ptr() if O is not None else None, L.data_ptr(),
                                      Oacc.data_ptr() if Oacc is not None else None,
                                      M.data_ptr() if M is not None else None,
                                      B, H, Nq, Nk, d, float(softmax_scale), _dtype_code(Q),
                                      1 if first else 0, 1 if last else 0, _stream_ptr(stream))
    check(st, "fa2_forward_step")


class _Attention(torch.autograd.Function):
    """flash_attention_2_forward / _backward as one differentiable op (device tensors in, device tensors out; the kernels
    are the only arithmetic).  Saves Q, K, V, O and the log-sum-exp L, as the reference's backward expects them
    (02_flash_attention_v2_backward/flash_attention_backward_kernel.cu:249-262)."""

    @staticmethod
    def forward(ctx, Q, K, V, softmax_scale, causal):
        Q, K, V = Q.contiguous(), K.contiguous(), V.contiguous()
        scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(Q.shape[-1])
        O, L = flash_attention_2_forward(Q, K, V, scale, causal=causal)
        ctx.save_for_backward(Q, K, V, O, L)
        ctx.scale, ctx.causal = scale, bool(causal)
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, O, L = ctx.saved_tensors
        dQ, dK, dV = flash_attention_2_backward(Q, K, V, O, L, dO.contiguous(), ctx.scale, causal=ctx.causal)
        return dQ, dK, dV, None, None


def attention(Q, K, V, softmax_scale=None, causal=False):
    """softmax(scale Q K^T [causal]) V for [B, H, N, d] bf16 (d = 64 | 128) or fp32 (non-causal) device tensors, with gradients.
    A convenience for callers that live in torch autograd; tests and bench.py call the two halves directly."""
    return _Attention.apply(Q, K, V, softmax_scale, causal)


def read_clocks(stream=None):
    """One sample of fa2_read_clocks on `stream`: an int64 device tensor [16][2] = per XCC (shader-clock ticks, 100 MHz
    reference ticks); rows of XCCs the device does not have stay zero.  Asynchronous: synchronise before reading."""
    out = torch.zeros(16, 2, dtype=torch.int64, device="cuda")
    check(_capi.lib().fa2_read_clocks(out.data_ptr(), _stream_ptr(stream)), "fa2_read_clocks")
    return out


def mean_shader_clock_mhz(before, after):
    """Mean shader clock between two read_clocks samples (the caller has synchronised): per XCC present in both,
    d(ticks) / d(reference ticks) x 100 MHz; the mean over those XCCs.  s_memtime is an XCC's own counter, so a difference
    is only ever taken within one XCC."""
    a, b = before.cpu().tolist(), after.cpu().tolist()
    vals = [(y[0] - x[0]) / (y[1] - x[1]) * 100.0 for x, y in zip(a, b) if x[1] and y[1] and y[1] > x[1]]
    if not vals:
        raise RuntimeError("fa2_read_clocks: no XCC present in both samples")
    return sum(vals) / len(vals)


def bare_mfma_tflops(seconds=0.15, workgroups=256):
    """What this device sustains on nothing but bf16 MFMAs on random operands (fa2_mfma_probe): TFLOP/s over about `seconds`
    of back-to-back launches after a ramp, and the mean shader clock it held.  A measurement aid for bench.py."""
    lib = _capi.lib()
    ops = (torch.rand(1024, 8, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(workgroups * 256, device="cuda")
    s = _stream_ptr()
    iters = 20000                                        # 4 x 20000 MFMAs per wave: ~1.4 ms per launch
    run = lambda: check(lib.fa2_mfma_probe(ops.data_ptr(), out.data_ptr(), iters, workgroups, s), "fa2_mfma_probe")
    for _ in range(30):                                  # ramp: the clock settles under load
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0 = read_clocks()
    e0.record()
    n = 0
    while True:
        for _ in range(10):
            run()
        n += 10
        e1.record()
        e1.synchronize()
        if e0.elapsed_time(e1) >= seconds * 1e3:
            break
    c1 = read_clocks()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = 4.0 * iters * workgroups * 4 * 32768.0 * n
    return flops / ms / 1e9, mean_shader_clock_mhz(c0, c1)
