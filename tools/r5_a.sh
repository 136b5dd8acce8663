#!/bin/bash
# round 5, call A: phase stamps of the current stem123 kernel + the GPU suite at HEAD
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/stamps_stem123.py ) > gpurun_out/r5_stamps_stem123_base.txt 2>&1
( timeout 1500 python -m pytest tests -q -m gpu -x --timeout=600 2>&1 | tail -15 ) > gpurun_out/r5_pytest_gpu_base.log 2>&1
tail -30 gpurun_out/r5_stamps_stem123_base.txt; tail -3 gpurun_out/r5_pytest_gpu_base.log
