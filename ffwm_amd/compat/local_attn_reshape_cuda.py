"""``local_attn_reshape_cuda`` -- same call shape as the pybind module built from
/root/reference/cuda/local_attn_reshape/local_attn_reshape_cuda.cc:5-28."""
from .. import ops


def forward(inputs, output, kernel_size):
    ops.local_attn_reshape_forward(inputs, kernel_size, out=output)
    return 1


def backward(inputs, grad_output, grad_inputs, kernel_size):
    # reference semantics: += into the caller's (zero-filled) buffer
    # (a non-contiguous grad_output is read through its strides: ffwm_local_attn_reshape_backward_strided, ABI 5)
    ops.local_attn_reshape_backward(grad_output, kernel_size, grad_inputs, accumulate=True)
    return 1
