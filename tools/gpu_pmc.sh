#!/bin/bash
# PMC counters for one kernel (name pattern $1) of the forward pass (UBD_PMC_DTYPE=float32|bfloat16|float16, UBD_PMC_TRAIN=1: train step at batch 32; separate passes, kernel-trace only)
PAT=${1:-dilconv_wino}
TAG=$(echo "$PAT" | tr -c "A-Za-z0-9_\n" "_")
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt"
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
    lab = synthetic.rectangle_maps(30, 32, 128, 128)
    tx = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    tr = Trainer(m, Adam())
    for _ in range(2): tr.train_step_on_device(tx, torch.from_numpy(lab).cuda())
else:
    for _ in range(3): m.predict_on_device(x)
torch.cuda.synchronize()
PY
i=0
if [ -n "$UBD_PMC_ONLY" ]; then   # one custom counter set instead of the standard passes
  rm -rf /tmp/pmcx
  rocprofv3 --kernel-trace --pmc $UBD_PMC_ONLY --output-format csv -d /tmp/pmcx -- python3 /tmp/fwd_once.py > /dev/null 2>&1
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$PAT" <<'PY' >> "$OUT"
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v), "n=", len(v))
PY
  cat "$OUT"; exit 0
fi
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -- python3 /tmp/fwd_once.py > /dev/null 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$PAT" <<'PY' >> "$OUT"
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("ERR", e)
for k, v in acc.items():
    print(k, sum(v) / len(v), "n=", len(v))
PY
  i=$((i+1))
done
cat "$OUT"
