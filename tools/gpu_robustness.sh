#!/bin/bash
# Robustness evidence of a round, one gpurun call: randomized soaks with raised case counts, the bit-equality variant sweep, ten repeats of the
# whole GPU suite.  Logs land in gpurun_out/robust_*.log (copy to profiles/<round>_soak.log, _stress_variants.log, _gpu_suite_repeats.log).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== loss vs the fp64 oracle: 300 random cases (pixel counts, class counts, logit scales, label densities, quantised logits = ties)"
UBD_LOSS_SOAK_CASES=300 python3 -m pytest tests/test_gpu_loss.py -q -k soak 2>&1 | tail -1
echo "== fp32 forward vs the fp64 oracle: 120 random shapes (every stem variant, both paddings, uint8 cases)"
UBD_FWD_SOAK_CASES=120 python3 -m pytest tests/test_gpu_forward.py -q -k soak 2>&1 | tail -1
echo "== 16-bit forward vs the rounding-aware and the fp64 oracle: 80 random shapes"
UBD_FWD16_SOAK_CASES=80 python3 -m pytest tests/test_gpu_forward16.py -q -k soak 2>&1 | tail -1
echo "== postprocess vs the C oracle: 200 random maps"
UBD_PP_SOAK_CASES=200 python3 -m pytest tests/test_gpu_postprocess.py -q -k soak 2>&1 | tail -1
echo "== fp32 train step vs fp64 autograd: 100 random shapes (the round's new separable-backward kernels), then with the library sized for 1 and 3 CUs"
UBD_TRAIN_SOAK_CASES=100 python3 -m pytest tests/test_gpu_train.py -q -k soak 2>&1 | tail -1
UBD_TEST_NUM_CUS=1 UBD_TRAIN_SOAK_CASES=20 python3 -m pytest tests/test_gpu_train.py -q -k "soak or split or aligned" 2>&1 | tail -1
UBD_TEST_NUM_CUS=3 UBD_TRAIN_SOAK_CASES=20 python3 -m pytest tests/test_gpu_train.py -q -k "soak or split or aligned or bit_for_bit" 2>&1 | tail -1
} > gpurun_out/robust_soak.log 2>&1
{
echo "== kernel variants that must agree bit for bit, 40 random shapes + the fixed ones, 300 repeat launches at the headline sizes"
python3 tools/stress_variants.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/robust_stress_variants.log 2>&1
{
for i in 1 2 3 4 5 6 7 8 9 10; do echo "== full suite, run $i"; python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -1; done
} > gpurun_out/robust_gpu_suite_repeats.log 2>&1
tail -3 gpurun_out/robust_soak.log; tail -3 gpurun_out/robust_stress_variants.log; tail -4 gpurun_out/robust_gpu_suite_repeats.log
