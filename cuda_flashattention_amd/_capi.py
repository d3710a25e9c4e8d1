"""ctypes binding of libfa2_mi355x.so -- the reference-side stub a maintainer would write to
call the C ABI of include/fa2_mi355x.h.

The library is built in-tree (cuda_flashattention_amd/lib/) by cuda_flashattention_amd/csrc/
Makefile.  There is NO fallback: if the shared object is missing or a symbol cannot be
resolved, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA2_LIB_PATH: load another build of the SAME library (A/B runs of kernel variants); never a fallback.
LIB_PATH = os.environ.get("FA2_LIB_PATH") or os.path.join(_HERE, "lib", "libfa2_mi355x.so")
RING_LIB_PATH = os.path.join(_HERE, "lib", "libfa2_ring_mi355x.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fa2_mi355x.h")

FA2_OK = 0
FA2_DTYPE_BF16 = 0
FA2_DTYPE_F32 = 1
FA2_DTYPE_FP8_E4M3 = 2

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/fa2_mi355x.h declaration by declaration.
SIGNATURES = {
    "fa2_version": (ctypes.c_char_p, []),
    "fa2_status_string": (ctypes.c_char_p, [_i]),
    "flash_attention": (_i, [_vp] * 6 + [_i, _i, _i, _i]),
    "flash_attention_2_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f]),
    "flash_attention_2_backward": (_i, [_vp] * 9 + [_i, _i, _f]),
    "fa2_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "fa2_forward_fp8_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fa2_forward_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "fa2_forward_fp8_scaled": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _i, _vp, _sz, _vp]),
    "fa2_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "fa2_backward": (_i, [_vp] * 9 + [_i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "fa2_backward_plan": (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_char_p)]),
    "fa2_backward_status": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "fa2_backward_phases": (_i, [_vp] * 9 + [_i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp, _i]),
    "fa2_backward_fused_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fa2_backward_fused": (_i, [_vp] * 9 + [_i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "fa2_backward_block": (_i, [_vp] * 9 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i]),
    "fa2_forward_step": (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "fa2_forward_step_strided": (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fa2_forward_state_finalize": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "fa2_accumulate_bf16": (_i, [_vp, _vp, _sz, _i, _vp]),
    "fa2_accumulate_bf16_2d": (_i, [_vp, _vp, _sz, _sz, _sz, _i, _vp]),
    "fa2_read_clocks": (_i, [_vp, _vp]),
    "fa2_mfma_probe": (_i, [_vp, _vp, _i, _i, _vp]),
    "fa2_fill_f32": (_i, [_vp, _sz, _f, _vp]),
    "fa2_convert_f32_to_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "fa2_convert_bf16_to_f32": (_i, [_vp, _vp, _sz, _vp]),
}


class FA2Error(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        try:
            msg = lib().fa2_status_string(status).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__(f"{where}: status {status} ({msg})")


_lib = None


def lib():
    """The loaded library.  import torch first in a torch process so that the HIP runtime the
    library binds (libamdhip64.so.7 by SONAME) is the one torch already loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C cuda_flashattention_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, where):
    if status != FA2_OK:
        raise FA2Error(status, where)
