#!/bin/bash
# the 16-bit tests (forward, train, persistent kernels, BN, end to end) with the product library, then a same-box A/B of builds under tools/_ab/
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_forward16.py tests/test_gpu_train.py tests/test_gpu_persistent.py tests/test_gpu_bn.py tests/test_gpu_end_to_end.py -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8 )
( timeout 1200 python tools/ab_lib.py "$@" ) 2>&1 | cut -c1-110
