#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_forward16.py tests/test_gpu_train.py tests/test_gpu_persistent.py tests/test_gpu_comm.py tests/test_gpu_comm_loopback.py tests/test_gpu_end_to_end.py -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 ) > gpurun_out/r4g_pytest.log 2>&1
cat gpurun_out/r4g_pytest.log
( timeout 900 python tools/ab_lib.py new3.so@UBD_REDUCE=batched new3.so ) 2>&1 | cut -c1-220
