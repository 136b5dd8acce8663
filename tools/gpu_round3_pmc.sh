#!/bin/bash
# round 3: PMC counters of the two forward kernels and per-kernel HBM traffic (fp32 forward, bf16 train step)
bash tools/gpu_pmc.sh dilconv_wino > /dev/null 2>&1; cp gpurun_out/pmc_dilconv_wino.txt gpurun_out/r03_pmc_dilconv_wino.txt
bash tools/gpu_pmc.sh stem123_kernel > /dev/null 2>&1; cp gpurun_out/pmc_stem123_kernel.txt gpurun_out/r03_pmc_stem123_fp32.txt
bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic_float32.txt gpurun_out/r03_pmc_traffic_fwd_fp32.txt
UBD_PMC_DTYPE=bfloat16 UBD_PMC_TRAIN=1 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic_bfloat16_train.txt gpurun_out/r03_pmc_traffic_train_bf16.txt
head -30 gpurun_out/r03_pmc_dilconv_wino.txt; head -12 gpurun_out/r03_pmc_traffic_fwd_fp32.txt
