"""CPU: the Python mirror refuses what the C ABI (raw pointers) could only misread -- host tensors, non-contiguous
views (the normal autograd case for dO), mismatched shapes and dtypes -- before any pointer reaches the library."""
import pytest

torch = pytest.importorskip("torch")


class _Fake(torch.Tensor):
    """A CPU tensor that claims to live on a device: lets the validation logic run without a GPU."""
    @property
    def is_cuda(self):
        return True


def _t(*shape, dtype=torch.bfloat16):
    return torch.zeros(*shape, dtype=dtype).as_subclass(_Fake)


def test_forward_rejects_bad_tensors():
    from cuda_flashattention_amd import ops
    Q = _t(1, 2, 64, 128)
    with pytest.raises(ValueError, match="device tensor"):
        ops.flash_attention_2_forward(torch.zeros(1, 2, 64, 128, dtype=torch.bfloat16), Q, Q)
    with pytest.raises(ValueError, match="shape"):
        ops.flash_attention_2_forward(Q, _t(1, 2, 32, 128), Q)
    with pytest.raises(ValueError, match="dtype"):
        ops.flash_attention_2_forward(Q, _t(1, 2, 64, 128, dtype=torch.float32), Q)
    with pytest.raises(ValueError, match="dtype"):          # O for bf16 inputs must be bf16
        ops.flash_attention_2_forward(Q, Q, Q, O=_t(1, 2, 64, 128, dtype=torch.float32))
    with pytest.raises(ValueError, match="L"):
        ops.flash_attention_2_forward(Q, Q, Q, O=_t(1, 2, 64, 128), L=_t(1, 2, 63, dtype=torch.float32))


def test_backward_rejects_noncontiguous_dO_and_mismatched_O():
    from cuda_flashattention_amd import ops
    Q = _t(1, 2, 64, 128)
    L = _t(1, 2, 64, dtype=torch.float32)
    dO_view = _t(1, 2, 128, 64).transpose(2, 3)             # right shape, wrong strides: what autograd may hand over
    assert dO_view.shape == Q.shape and not dO_view.is_contiguous()
    with pytest.raises(ValueError, match="contiguous"):
        ops.flash_attention_2_backward(Q, Q, Q, Q, L, dO_view)
    with pytest.raises(ValueError, match="contiguous"):     # an expanded (stride-0) gradient
        ops.flash_attention_2_backward(Q, Q, Q, Q, L, _t(1, 1, 1, 1).expand(1, 2, 64, 128))
    with pytest.raises(ValueError, match="O"):
        ops.flash_attention_2_backward(Q, Q, Q, _t(1, 2, 64, 64), L, Q)
    with pytest.raises(ValueError, match="dK"):
        ops.flash_attention_2_backward(Q, Q, Q, Q, L, Q, dK=_t(1, 2, 32, 128))
    with pytest.raises(ValueError, match="workspace"):
        ops.flash_attention_2_backward(Q, Q, Q, Q, L, Q, workspace=torch.zeros(16, dtype=torch.uint8))
