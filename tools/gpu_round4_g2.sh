#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -8 ) > gpurun_out/r4g_pytest.log 2>&1
cat gpurun_out/r4g_pytest.log
