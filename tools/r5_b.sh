#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tools/ab_stem.py ) > gpurun_out/r5_ab_stem.txt 2>&1
( timeout 600 python tools/stamps_stem123w.py ) > gpurun_out/r5_stamps_stem123w.txt 2>&1
cat gpurun_out/r5_ab_stem.txt; tail -32 gpurun_out/r5_stamps_stem123w.txt
