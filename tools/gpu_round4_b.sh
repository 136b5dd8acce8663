#!/bin/bash
# round 4, second pass: the new parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_bn.py tests/test_gpu_comm_loopback.py tests/test_gpu_raster.py tests/test_gpu_forward.py tests/test_gpu_postprocess.py -q -m gpu -x --timeout=900 -s 2>&1 | grep -v "^$" | tail -25 ) > gpurun_out/r4b_pytest_gpu.log 2>&1
tail -12 gpurun_out/r4b_pytest_gpu.log
