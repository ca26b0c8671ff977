#!/bin/bash
# usage (GPU box): scripts/pmc_fetch.sh <tag> <python script + args...> ; separate --pmc passes for FETCH_SIZE and WRITE_SIZE, per kernel+grid means
tag=$1; shift
# the passes run from /tmp: make a repo-relative script path absolute
case "$1" in /*) ;; *) set -- "$GRAFT_REPO_ROOT/$1" "${@:2}" ;; esac
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/$tag.$c -o r --output-format csv -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.$c.log 2>&1
done
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
import csv, glob, collections
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python $@")
print("# counter values are KiB per dispatch (mean); FETCH_SIZE counts 128-byte requests as 64 on gfx950: double it (MI355X_MICROARCH.md)")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag.%s/**/*counter_collection.csv" % c, recursive=True)
    vals = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: vals[(r["Kernel_Name"][:64], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items()):
        if any(s in k[0] for s in ("gemm", "attn", "epilogue", "lora", "class_areas", "mask_counts", "fmeasure_hist")):
            print(f"{c}: {k[0]} grid {k[1]}: {len(v)} dispatches, mean {sum(v)/len(v):.6g} KiB")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
