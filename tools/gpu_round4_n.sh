#!/bin/bash
# tests of the bf16 paths with the product library, then a same-box A/B of builds under tools/_ab/
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_forward16.py tests/test_gpu_persistent.py -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8 )
( timeout 1200 python tools/ab_lib.py "$@" ) 2>&1 | cut -c1-150
