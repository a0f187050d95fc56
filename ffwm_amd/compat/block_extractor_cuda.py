"""``block_extractor_cuda`` -- same call shape as the pybind module built from
/root/reference/cuda/block_extractor/block_extractor_cuda.cc:5-32."""
from .. import ops


def forward(source, flow_field, output, kernel_size):
    ops.block_extractor_forward(source, flow_field, kernel_size, out=output)
    return 1


def backward(source, flow_field, grad_output, grad_source, grad_flow_field, kernel_size):
    # the reference kernels read grad_output through its strides (block_extractor_kernel.cu:8-15); so does the C ABI since ABI 5
    # (ffwm_block_extractor_backward_strided): no copy here
    ops.block_extractor_backward(source, flow_field, grad_output, kernel_size, grad_source, grad_flow_field)
    return 1
