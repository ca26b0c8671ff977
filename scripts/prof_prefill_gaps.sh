#!/bin/bash
# usage (on the GPU box): scripts/prof_prefill_gaps.sh <tag> [B]   -> gpurun_out/<tag>.txt: the prefill phase (encoders + splice + chunked decoder prefill of B clips,
# scripts/probe_prefill.py, chunks of 35) under rocprofv3 --kernel-trace: busy / idle time of the LAST timed pass with the kernel pairs that own the idle
# time, and the per-kernel table of the same window
tag=$1; B=${2:-448}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/scripts/probe_prefill.py $B 1 both 35 > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
{
  grep -E "ms/clip|phase" $GRAFT_REPO_ROOT/gpurun_out/$tag.log
  python - <<PY
import csv, glob, collections, re
f = [x for x in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/**/*.csv", recursive=True) if "kernel_trace" in x][0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# the probe runs enc + pre once (warm-up) and once timed: the timed pass = the second half of the dispatches
half = len(ev) // 2
sel = ev[half:]
span = sel[-1][1] - sel[0][0]
busy, cur, prev = 0, sel[0][0], None
gaps, gapn = collections.Counter(), collections.Counter()
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:44]
for s, e, n in sel:
    if s > cur and prev is not None:
        k = short(prev) + " -> " + short(n)
        gaps[k] += s - cur; gapn[k] += 1
    busy += max(0, e - max(s, cur))
    if e > cur: cur, prev = e, n
print(f"## timed pass: {span/1e6:.1f} ms wall, busy {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms ({100*(span-busy)/span:.1f} %), {len(sel)} kernels")
print("## idle time by (previous kernel -> next kernel), top 25")
for k, v in gaps.most_common(25):
    print(f"{v/1e6:8.2f} ms {gapn[k]:6d} x {v/gapn[k]/1e3:8.1f} us  {k}")
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in sel:
    k = short(n)[:70]; tot[k] += e - s; cnt[k] += 1
print("## per-kernel durations inside the timed pass")
for k, v in tot.most_common(28):
    print(f"{v/1e6:9.2f} ms {cnt[k]:6d} x {v/cnt[k]/1e3:8.2f} us  {100*v/span:5.1f}%  {k}")
PY
} > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
