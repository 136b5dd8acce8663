#!/bin/bash
# the two committed bench lines (default flags; the driver's flags) on the committed profile set
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python bench.py ) > gpurun_out/r04_bench.log 2> gpurun_out/r04_bench.err; grep '^{' gpurun_out/r04_bench.log | tail -1 > gpurun_out/r04_bench_line.json
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04_bench_driver_flags.log 2>/dev/null; grep '^{' gpurun_out/r04_bench_driver_flags.log | tail -1 > gpurun_out/r04_bench_line_driver_flags.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_line.json')); d2=json.load(open('gpurun_out/r04_bench_line_driver_flags.json'))
print(d['value'], d['ms_per_step'], d['train_step']['ms_per_step'], d['forward_fp16_cfg5']['ms_per_batch'], d['roofline']['from_committed_profile']['taken_on_the_sources_of_this_build'])
print(d2['value'], d2['ms_per_step'], d2['train_step']['ms_per_step'], d2['forward_fp16_cfg5']['ms_per_batch'])
PY
