#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, kernel-trace only) of EVERY kernel of one pass:
# UBD_PMC_DTYPE=float32|bfloat16|float16, UBD_PMC_TRAIN=1 for a train step at batch 64 (forward passes: batch 32).  Table -> gpurun_out/pmc_traffic_<tag>.txt
TAG=${UBD_PMC_DTYPE:-float32}${UBD_PMC_TRAIN:+_train}
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_${TAG}.txt"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; rm -f "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fwd_once.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
m = Model(NetConfig(grey=False), dtype=os.environ.get("UBD_PMC_DTYPE", "float32"), seed=1)
x = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
if os.environ.get("UBD_PMC_TRAIN"):
    import numpy as np
    from ubdvss_amd import Trainer, Adam
    lab = synthetic.rectangle_maps(30, 64, 128, 128)      # the train configuration: 64 images per GPU
    tx = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    tr = Trainer(m, Adam())
    for _ in range(2): tr.train_step_on_device(tx, torch.from_numpy(lab).cuda())
else:
    for _ in range(3): m.predict_on_device(x)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 /tmp/fwd_once.py > /dev/null 2>&1
done
python3 - "$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" <<'PY' > "$OUT"
import csv, sys, collections
acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for f in sys.argv[1:3]:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
import os
print("# per launch, batch " + ("64 (the train configuration)" if os.environ.get("UBD_PMC_TRAIN") else "32") + " of 512x512x3; FETCH_SIZE doubled (gfx950 64-byte units, MI355X_MICROARCH.md), KiB -> MB")
print(f"{'kernel':70s} {'launches':>8s} {'read MB':>9s} {'write MB':>9s}")
names = sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]), key=lambda k: -sum(acc["FETCH_SIZE"].get(k, [0])))
for k in names:
    fv, wv = acc["FETCH_SIZE"].get(k, [0.0]), acc["WRITE_SIZE"].get(k, [0.0])
    rd = 2.0 * sum(fv) / len(fv) * 1024 / 1e6
    wr = sum(wv) / len(wv) * 1024 / 1e6
    if rd + wr < 0.5: continue
    print(f"{k[:70]:70s} {len(fv):8d} {rd:9.1f} {wr:9.1f}")
PY
cat "$OUT"
