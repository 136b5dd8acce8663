"""Copies the outputs of tools/gpu_profile_set.sh from gpurun_out/ into profiles/<round>_*: the bench line = the MEDIAN (by value) of the
three default runs, the line measured with the driver's flags, kernel stats, traffic tables; then writes the profile meta (fingerprint of the
kernel sources).  usage: python tools/collect_profiles.py r05"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
lines = []
for i in (1, 2, 3):
    txt = open(os.path.join(G, f"bench_default_{i}.json")).read().strip().splitlines()
    lines.append(json.loads(txt[-1]))
lines.sort(key=lambda d: d["value"])
print("default runs:", [d["value"] for d in lines])
json.dump(lines[1], open(os.path.join(P, f"{rnd}_bench_line.json"), "w"), indent=1)
json.dump(json.loads(open(os.path.join(G, "bench_driver_flags.json")).read().strip().splitlines()[-1]), open(os.path.join(P, f"{rnd}_bench_line_driver_flags.json"), "w"), indent=1)
for src, dst in (("bench_kernel_stats.csv", "bench_kernel_stats.csv"), ("bench_full_kernel_stats.csv", "bench_full_kernel_stats.csv"),
                 ("pmc_traffic_float32.txt", "pmc_traffic_fwd_fp32.txt"), ("pmc_traffic_float16.txt", "pmc_traffic_fwd_fp16.txt"),
                 ("pmc_traffic_bfloat16_train.txt", "pmc_traffic_train_bf16.txt"), ("pmc_traffic_float32_train.txt", "pmc_traffic_train_fp32.txt")):
    if os.path.exists(os.path.join(G, src)): shutil.copy(os.path.join(G, src), os.path.join(P, f"{rnd}_{dst}"))
    else: print("missing", src)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "profile_meta.py"), rnd])
