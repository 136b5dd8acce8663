"""Observed 16-bit parity figures on the MI355X (run on the GPU box): every forward case of tests/test_gpu_forward16.py
plus the bf16/fp16 gradient errors, written to gpurun_out/parity16.json.  The test gates are 3 x these maxima."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_forward16 as t

cases = [(3, 0, True, 2, 128, 128, 53), (1, 3, True, 1, 64, 192, 54), (3, 2, False, 2, 72, 100, 55), (3, 0, True, 2, 512, 512, 60)]
rep = {}
for dtype in ("bfloat16", "float16"):
    rows = [dict(case=c, **t.measure(dtype, *c)) for c in cases]
    if dtype == "float16":
        rows.append(dict(case=(3, 0, True, 2, 1024, 1024, 77), **t.measure(dtype, 3, 0, True, 2, 1024, 1024, 77)))
    rep[dtype] = {"cases": rows, "max_e16": max(r["e16"] for r in rows), "max_e64": max(r["e64"] for r in rows)}
    print(dtype, "max e16 %.3e  max e64 %.3e" % (rep[dtype]["max_e16"], rep[dtype]["max_e64"]))
    for r in rows:
        print("  ", r["case"], "e16 %.2e e64 %.2e" % (r["e16"], r["e64"]), r["map_same_rounding"], r["map_fp64"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "parity16.json"), "w"), indent=1)
