#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/ab_lib.py r3.so new3.so ) 2>&1 | cut -c1-220
( timeout 900 python bench.py --no-cpu-baseline --steps 200 ) 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench: step',d['ms_per_step'],'train',d['train_step']['ms_per_step'],d['train_step']['ms_per_step_single_tensor'],'cfg5',d['forward_fp16_cfg5']['ms_per_batch'],d['forward_fp16_cfg5']['ms_per_batch_single_tensor'])"
