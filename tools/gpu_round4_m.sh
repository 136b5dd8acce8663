#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( UBD_LIB_PATH=$PWD/tools/_ab/$1 timeout 1800 python -m pytest tests/test_gpu_forward16.py tests/test_gpu_train.py -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -5 )
shift
( timeout 900 python tools/ab_lib.py "$@" ) 2>&1 | cut -c1-120
