#!/bin/bash
# round 5: evidence for the stem re-architecture experiment (profiles/r05_experiment_notes.txt)
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/_ab/l2unit > gpurun_out/r05_ubench_l2unit.txt 2>&1
tools/_ab/mfma_fill > gpurun_out/r05_ubench_mfma_fill.txt 2>&1
( timeout 600 python tools/stamps_stem123.py ) > gpurun_out/r05_stamps_stem123.txt 2>&1
( timeout 600 python tools/stamps_stem123w.py ) > gpurun_out/r05_stamps_stem123w.txt 2>&1
( AB_ROUNDS=3 timeout 900 python tools/ab_stem.py ) > gpurun_out/r05_ab_stem.txt 2>&1
tail -8 gpurun_out/r05_ab_stem.txt
