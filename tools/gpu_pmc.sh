#!/bin/bash
# PMC counters for the dilated-conv kernel (separate passes, kernel-trace only as gpurun requires)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/bench_layer.py --reps 20 > $GRAFT_REPO_ROOT/gpurun_out/layer_bench.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -- python3 $GRAFT_REPO_ROOT/tools/bench_layer.py --reps 3 --layers 2 > /dev/null 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/pmc_dilconv.txt
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "dilconv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("ERR", e)
for k, v in acc.items():
    print(k, sum(v) / len(v), "n=", len(v))
PY
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/layer_bench.log; cat gpurun_out/pmc_dilconv.txt
