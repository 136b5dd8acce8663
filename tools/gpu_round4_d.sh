#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
LINES_OUT=12 bash tools/gpu_prof_py.sh tools/bench_cfg5.py; cp gpurun_out/kstats_bench_cfg5.csv gpurun_out/r4d_kstats_cfg5_fused123.csv
export UBD_STEM16=fused12
LINES_OUT=12 bash tools/gpu_prof_py.sh tools/bench_cfg5.py; cp gpurun_out/kstats_bench_cfg5.csv gpurun_out/r4d_kstats_cfg5_fused12.csv
