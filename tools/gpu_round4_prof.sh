#!/bin/bash
# round 4: the committed evidence -- smoke, default bench line, rocprofv3 kernel stats (forward legs; all legs), PMC traffic tables, PMC counters
# of the two dominant forward kernels.  Everything lands in gpurun_out/ as r04_* (copied to profiles/ by hand).
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r04_smoke.log 2>&1; tail -1 gpurun_out/r04_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r04_bench.log 2> gpurun_out/r04_bench.err; grep '^{' gpurun_out/r04_bench.log | tail -1 > gpurun_out/r04_bench_line.json; tail -c 300 gpurun_out/r04_bench_line.json; tail -2 gpurun_out/r04_bench.err
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04_bench_driver_flags.log 2>/dev/null; grep '^{' gpurun_out/r04_bench_driver_flags.log | tail -1 > gpurun_out/r04_bench_line_driver_flags.json
cd /tmp
rm -rf /tmp/prof_fwd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/gpurun_out/r04_prof_bench_fwd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/full_size_avg.py $(find /tmp/prof_fwd -name "*kernel_trace.csv" | head -1) $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_bench_kernel_stats.csv
rm -rf /tmp/prof_full && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_prof_bench_full.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/full_size_avg.py $(find /tmp/prof_full -name "*kernel_trace.csv" | head -1) $(find /tmp/prof_full -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_bench_full_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cut -c1-150 gpurun_out/r04_bench_kernel_stats.csv | head -12
bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic_float32.txt gpurun_out/r04_pmc_traffic_fwd_fp32.txt
UBD_PMC_DTYPE=bfloat16 UBD_PMC_TRAIN=1 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic_bfloat16_train.txt gpurun_out/r04_pmc_traffic_train_bf16.txt
UBD_PMC_DTYPE=float16 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic_float16.txt gpurun_out/r04_pmc_traffic_fwd_fp16.txt
bash tools/gpu_pmc.sh dilconv_wino > /dev/null 2>&1; cp gpurun_out/pmc_dilconv_wino.txt gpurun_out/r04_pmc_dilconv_wino.txt
bash tools/gpu_pmc.sh stem123_kernel > /dev/null 2>&1; cp gpurun_out/pmc_stem123_kernel.txt gpurun_out/r04_pmc_stem123_fp32.txt
head -14 gpurun_out/r04_pmc_traffic_train_bf16.txt
UBD_PMC_DTYPE=float16 bash tools/gpu_pmc.sh dilconv16s > /dev/null 2>&1; cp gpurun_out/pmc_dilconv16s.txt gpurun_out/r04_pmc_dilconv16s_fp16.txt
UBD_PMC_DTYPE=float16 bash tools/gpu_pmc.sh sep123_16 > /dev/null 2>&1; cp gpurun_out/pmc_sep123_16.txt gpurun_out/r04_pmc_sep123_16_fp16.txt
grep -E "BANK_CONFLICT|IDX_ACTIVE" gpurun_out/r04_pmc_sep123_16_fp16.txt
