#!/bin/bash
# usage (GPU box): scripts/pmc_gemm.sh <tag> <M> <tunes...> ; PMC pass over bench_gemm.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_gemm.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/$tag | head
python - <<PY
import csv, glob, collections
fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/**/*counter_collection.csv", recursive=True)
print(fs)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "gemm" in k[0]:
        print(k, {n: f"{v:.3g}" for n, v in d.items()})
PY
