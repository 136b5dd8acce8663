#!/bin/bash
# round 4: the whole -m gpu suite ten times back to back (a fresh process each), then the postprocess stress tests at 3000
# launches per case, with the library sized for a one-CU device, and the persistent-kernel variants sweep; log -> profiles/
mkdir -p gpurun_out
LOG=gpurun_out/r04_gpu_suite_repeats.log
: > $LOG
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "== full suite, run $i" >> $LOG
  ( timeout 1200 python -m pytest tests -q -m gpu -x --timeout=900 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|error" | tail -2 ) >> $LOG 2>&1
done
echo "== postprocess stress, 3000 launches per case" >> $LOG
( UBD_PP_STRESS_LAUNCHES=3000 timeout 1500 python -m pytest tests/test_gpu_postprocess.py -q -m gpu -k "stress" 2>&1 | grep -a "passed\|failed" | tail -2 ) >> $LOG 2>&1
echo "== postprocess tests with the library sized for ONE CU (UBD_TEST_NUM_CUS=1)" >> $LOG
( UBD_TEST_NUM_CUS=1 timeout 1500 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_forward.py -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2 ) >> $LOG 2>&1
echo "== kernel variants on ragged shapes (tools/stress_variants.py)" >> $LOG
( timeout 1500 python tools/stress_variants.py 2>&1 | tail -6 ) >> $LOG 2>&1
cat $LOG
