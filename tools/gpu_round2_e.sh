#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_persistent.py tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/stamps_wino.py 2>&1 | grep -v amdgpu.ids | head -5
timeout 300 python tools/bench_layer.py 2>&1 | grep -v amdgpu.ids | tail -12
