#!/bin/bash
# usage (GPU box): scripts/pmc_valu.sh <tag> <python script + args...> ; one --pmc pass of SQ issue counters (VALU / MFMA / LDS / VMEM activity, barrier waits), per kernel+grid sums
tag=$1; shift
case "$1" in /*) ;; *) set -- "$GRAFT_REPO_ROOT/$1" "${@:2}" ;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r --output-format csv -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
import csv, glob, collections
fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:64], r.get("Grid_Size", ""))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
print("# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -- python $@")
for k, d in sorted(agg.items()):
    if any(s in k[0] for s in ("gemm", "attn", "epilogue", "lora")) and d.get("SQ_WAVE_CYCLES"):
        print(f"{k[0]} grid {k[1]} ({n[k]} dispatches): " + " ".join(f"{c}={v / n[k]:.4g}" for c, v in sorted(d.items())))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
