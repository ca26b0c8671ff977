#!/bin/bash
# usage (on the GPU box): scripts/prof_decode_gaps.sh <tag> <B> <NEW> [llm]   -> gpurun_out/<tag>.txt: per-kernel table of the decode probe and, from the
# kernel trace's time stamps, busy / idle time of the last 40 ms (pure graph replay) with the kernel pairs that own the idle time
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/scripts/probe_decode.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
{
  grep -E "ms/step" $GRAFT_REPO_ROOT/gpurun_out/$tag.log
  echo "## busy / idle over the last 40 ms of the trace (graph replay of decode steps), idle time by (previous kernel -> next kernel)"
  python $GRAFT_REPO_ROOT/scripts/gap_report.py $GRAFT_REPO_ROOT/gpurun_out/$tag 40 0
  echo "## per-kernel durations inside the same window"
  python - <<PY
import csv, glob, collections
f = [x for x in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/**/*.csv", recursive=True) if "kernel_trace" in x][0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
end = ev[-1][1]
sel = [e for e in ev if e[0] >= end - 40e6]
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in sel:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    tot[k] += e - s; cnt[k] += 1
span = sel[-1][1] - sel[0][0]
for k, v in tot.most_common(16):
    print(f"{v/1e6:8.3f} ms {cnt[k]:6d} x {v/cnt[k]/1e3:7.2f} us  {100*v/span:5.1f}%  {k}")
PY
} > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
