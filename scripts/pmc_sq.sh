#!/bin/bash
# usage (GPU box): scripts/pmc_sq.sh <tag> <absolute python script + args...> ; one --pmc pass of SQ LDS / MFMA / wait counters, per kernel+grid sums
tag=$1; shift
# the passes run from /tmp: make a repo-relative script path absolute
case "$1" in /*) ;; *) set -- "$GRAFT_REPO_ROOT/$1" "${@:2}" ;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r --output-format csv -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
import csv, glob, collections
fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in fs:
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:64], r.get("Grid_Size", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
print("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS -- python $@")
for k, d in sorted(agg.items()):
    if any(s in k[0] for s in ("gemm", "attn", "epilogue", "lora")) and d.get("SQ_BUSY_CU_CYCLES"):
        b = d["SQ_BUSY_CU_CYCLES"]
        print(f"{k[0]} grid {k[1]}: MFMA busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / b / 4 * 100:.1f} %  LDS array {d['SQ_LDS_IDX_ACTIVE'] / b * 100:.1f} % of CU cycles  "
              f"wait any {d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1) * 100:.1f} %  wait LDS {d['SQ_WAIT_INST_LDS'] / max(d['SQ_WAVE_CYCLES'], 1) * 100:.1f} %  "
              f"conflicts {d['SQ_LDS_BANK_CONFLICT']:.3g}  | " + " ".join(f"{n}={v:.3g}" for n, v in sorted(d.items())))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
