#!/bin/bash
# round 4, final library: long randomized soaks of the oracle comparisons and of the bit-identical kernel variants; log -> profiles/r04_soak.log
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/r04_soak.log
: > $LOG
echo "== loss vs the fp64 oracle: 300 random cases (pixel counts, class counts, logit scales, label densities, quantised logits = ties)" >> $LOG
( UBD_LOSS_SOAK_CASES=300 timeout 1500 python -m pytest tests/test_gpu_loss.py -q -m gpu -k soak 2>&1 | grep -a "passed\|failed" | tail -1 ) >> $LOG 2>&1
echo "== 16-bit forward vs the rounding-aware and the fp64 oracle: 80 random shapes" >> $LOG
( UBD_FWD16_SOAK_CASES=80 timeout 2400 python -m pytest tests/test_gpu_forward16.py -q -m gpu -k "random_shape_soak" 2>&1 | grep -a "passed\|failed" | tail -1 ) >> $LOG 2>&1
echo "== kernel variants that must agree bit for bit, 40 random shapes + the fixed ones, 300 repeat launches at the headline sizes" >> $LOG
( UBD_VARIANT_RANDOM_SHAPES=40 UBD_VARIANT_REPEATS=300 timeout 3000 python tools/stress_variants.py 2>&1 | grep -v ": True" | tail -6 ) >> $LOG 2>&1
cat $LOG
