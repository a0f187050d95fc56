set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv2d_forward_mfma or block_extractor or block_attention or conv_forward_routing" 2>&1 | tail -15 > gpurun_out/r06_t1.log
python tools/r06/be_flush_ab.py > gpurun_out/r06_be_flush_ab.txt 2>&1
tail -30 gpurun_out/r06_t1.log; cat gpurun_out/r06_be_flush_ab.txt | grep -v amdgpu.ids
