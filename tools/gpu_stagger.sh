for us in 0 3 6 10 15; do UBD_STAGGER_US=$us timeout 300 python bench.py --no-cpu-baseline --no-train --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger', $us, d['value'], d['ms_per_step'], d['ms_per_step_spread']['median'], d['parts']['net_ms'])"; done
UBD_STAGGER_US=6 bash tools/gpu_timeline.sh 2>&1 | tail -8
