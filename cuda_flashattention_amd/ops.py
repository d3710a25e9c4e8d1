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
    if not x.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if not x.is_contiguous():
        raise ValueError(f"{name} must be contiguous [B][H][N][d]")
    if x.dim() == 2:
        return 1, 1, x.shape[0], x.shape[1]
    if x.dim() == 4:
        return tuple(x.shape)
    raise ValueError(f"{name}: expected [N,d] or [B,H,N,d], got {tuple(x.shape)}")


def flash_attention_2_forward(Q, K, V, softmax_scale=None, causal=False, O=None, L=None, stream=None):
    """O, L = FA2 forward.  Mirrors flash_attention_2_forward(Q,K,V,O,L,seq_len,head_dim,scale)
    (reference 02_forward/flash_attention_kernel.cu:300-309) with B,H,dtype,causal,stream added.
    Tensors [N,d] or [B,H,N,d]; L is fp32 [.., N] natural-log LSE."""
    B, H, N, d = _bhnd(Q, "Q")
    for n, t in (("K", K), ("V", V)):
        if _bhnd(t, n) != (B, H, N, d) or t.dtype != Q.dtype:
            raise ValueError(f"{n} must match Q in shape and dtype")
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    if O is None:     # fp8 inputs (OCP e4m3, d = 128, BASELINE configs[4]) produce a bf16 O
        O = torch.empty(Q.shape, dtype=torch.bfloat16, device=Q.device) if Q.dtype == torch.float8_e4m3fn else torch.empty_like(Q)
    if L is None:
        L = torch.empty(Q.shape[:-1], dtype=torch.float32, device=Q.device)
    st = _capi.lib().fa2_forward(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
                                 B, H, N, d, scale, _dtype_code(Q), 1 if causal else 0, _stream_ptr(stream))
    check(st, "fa2_forward")
    return O, L


def flash_attention_2_backward(Q, K, V, O, L, dO, softmax_scale=None, causal=False,
                               dQ=None, dK=None, dV=None, workspace=None, stream=None, phases=7):
    """dQ, dK, dV = FA2 backward.  Mirrors flash_attention_2_backward(Q,K,V,O,L,dO,dQ,dK,dV,...)
    (reference 02_backward/flash_attention_backward_kernel.cu:249-262)."""
    B, H, N, d = _bhnd(Q, "Q")
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(d)
    dQ = torch.empty_like(Q) if dQ is None else dQ
    dK = torch.empty_like(K) if dK is None else dK
    dV = torch.empty_like(V) if dV is None else dV
    lib = _capi.lib()
    need = lib.fa2_backward_workspace_bytes(B, H, N, d, _dtype_code(Q))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=Q.device)
    st = lib.fa2_backward_phases(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
                                 dO.data_ptr(), dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                                 B, H, N, d, scale, _dtype_code(Q), 1 if causal else 0,
                                 workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                 _stream_ptr(stream), int(phases))
    check(st, "fa2_backward")
    return dQ, dK, dV


def forward_step(Q, K, V, O, L, Oacc, M, softmax_scale, first, last, stream=None):
    """One resumable ring step (ring_attention_forward_kernel, ring_attention_kernel.cu:13-140)."""
    B, H, Nq, d = _bhnd(Q, "Q")
    _, _, Nk, _ = _bhnd(K, "K")
    st = _capi.lib().fa2_forward_step(Q.data_ptr(), K.data_ptr(), V.data_ptr(),
                                      O.data_ptr() if O is not None else None, L.data_ptr(),
                                      Oacc.data_ptr() if Oacc is not None else None,
                                      M.data_ptr() if M is not None else None,
                                      B, H, Nq, Nk, d, float(softmax_scale), _dtype_code(Q),
                                      1 if first else 0, 1 if last else 0, _stream_ptr(stream))
    check(st, "fa2_forward_step")
