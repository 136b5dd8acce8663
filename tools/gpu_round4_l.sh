#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8 )
