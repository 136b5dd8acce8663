#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_forward16.py tests/test_gpu_persistent.py -q -m gpu -x --timeout=900 2>&1 | tail -5 ) > gpurun_out/r4e_pytest.log 2>&1
tail -3 gpurun_out/r4e_pytest.log
( timeout 900 python tools/ab_lib.py r3.so new.so ) 2>&1 | cut -c1-220
