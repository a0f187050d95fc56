"""``resample2d_cuda`` -- same call shape as the pybind module built from
/root/reference/cuda/resample2d_package/resample2d_cuda.cc:6-32."""
from .. import ops


def forward(input1, input2, output, kernel_size, dilation):
    ops.resample2d_forward(input1, input2, kernel_size, dilation, out=output)
    return 1


def backward(input1, input2, gradOutput, gradInput1, gradInput2, kernel_size, dilation):
    # (a non-contiguous gradOutput is read through its strides: ffwm_resample2d_backward_strided, ABI 5)
    ops.resample2d_backward(input1, input2, gradOutput, kernel_size, dilation, gradInput1, gradInput2, reference_quirk=True)
    return 1
