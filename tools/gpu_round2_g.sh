# stem / pipeline check: fused-stem parity tests, the forward line at two stagger settings, the timeline of the pipelined step
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_end_to_end.py -m gpu -x -q 2>&1 | tail -3
for us in 0 3; do UBD_STAGGER_US=$us timeout 300 python bench.py --no-cpu-baseline --no-train --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('latency_batch1',{}); print('stagger', $us, d['value'], d['ms_per_step'], d['parts']['net_ms'], d['roofline_forward_pass']['frac'], [ (k, v.get('device_ms')) for k,v in l.items() if isinstance(v, dict)])"; done
bash tools/gpu_timeline.sh 2>&1 | tail -22
