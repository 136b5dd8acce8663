"""Writes profiles/<round>_profile_meta.json: the fingerprint of the kernel sources the committed profile set of that round was
measured on (bench.csrc_sha16).  bench.py compares it with the sources of the running build and says on its line whether the
profile-derived fields (traffic, profile_avg_us, frac_profile) belong to this build.  Usage: python tools/profile_meta.py r04"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
meta = {"csrc_sha16": bench.csrc_sha16(), "written": time.strftime("%Y-%m-%d %H:%M:%S"),
        "files": sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith(rnd + "_"))}
json.dump(meta, open(os.path.join(ROOT, "profiles", f"{rnd}_profile_meta.json"), "w"), indent=1)
print(meta["csrc_sha16"])
