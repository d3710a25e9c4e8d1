"""MI355X-native FlashAttention-2 hot path (fwd / bwd / ring) behind the reference's own
host-wrapper interface.  HIP kernels + C ABI live in csrc/ (libfa2_mi355x.so); this package
is the thin host-side mirror used by tests and bench.py.  No CPU fallback exists."""
from . import _capi  # noqa: F401
from .ops import flash_attention_2_forward, flash_attention_2_backward, forward_step, attention  # noqa: F401
