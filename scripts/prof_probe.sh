#!/bin/bash
# usage (on the GPU box): scripts/prof_probe.sh <tag> <probe args...>   -> gpurun_out/<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/scripts/probe_perf.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/gpurun_out/$tag/r_results.db $GRAFT_REPO_ROOT/gpurun_out/$tag.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
grep -E "decode step|prefill 4|encoders" $GRAFT_REPO_ROOT/gpurun_out/$tag.log
