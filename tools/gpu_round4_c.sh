#!/bin/bash
# round 4: the one-kernel 16-bit stem: parity, then same-box A/B against the round-3 library
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_forward16.py tests/test_gpu_train.py tests/test_gpu_persistent.py tests/test_gpu_bn.py -q -m gpu -x --timeout=900 2>&1 | tail -15 ) > gpurun_out/r4c_pytest.log 2>&1
tail -8 gpurun_out/r4c_pytest.log
cp ubdvss_amd/libubd_hip.so tools/_ab/new.so
( timeout 900 python tools/ab_lib.py r3.so new.so new.so@UBD_STEM16=fused12 ) > gpurun_out/r4c_ab.log 2>&1; cat gpurun_out/r4c_ab.log
