#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -5 )
( timeout 900 python tools/ab_lib.py new3.so new4.so b_th8_4.so ) 2>&1 | cut -c1-120
