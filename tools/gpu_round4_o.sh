#!/bin/bash
# loss tests + train tests with the product library, per-kernel stats of the bf16 train step, same-box A/B of builds under tools/_ab/
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_loss.py tests/test_gpu_train.py -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8 )
bash tools/gpu_train_stats.sh 2>&1 | grep -E "loss_|train step"
( timeout 1200 python tools/ab_lib.py "$@" ) 2>&1 | cut -c1-110
