#!/bin/bash
# usage (GPU box): scripts/pmc_attn.sh <tag> [B] [ctx]; separate --pmc passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/$tag.$c -o r --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_attn_decode.py "$@" 5 > $GRAFT_REPO_ROOT/gpurun_out/$tag.$c.log 2>&1
done
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
import csv, glob, collections
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/bench_attn_decode.py $@")
print(open("$GRAFT_REPO_ROOT/gpurun_out/$tag.FETCH_SIZE.log").read().strip().splitlines()[-1])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag.%s/**/*counter_collection.csv" % c, recursive=True)
    vals = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: vals[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        if "attn_decode" in k:
            print(f"{c}: kernel {k}: {len(v)} dispatches, mean counter value {sum(v)/len(v):.6g} (KiB)")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
