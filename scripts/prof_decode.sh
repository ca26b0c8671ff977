#!/bin/bash
# usage (on the GPU box): scripts/prof_decode.sh <tag> <B> <NEW> [llm]   -> gpurun_out/<tag>.txt (+ .log)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/scripts/probe_decode.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/gpurun_out/$tag/r_results.db $GRAFT_REPO_ROOT/gpurun_out/$tag.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
grep -E "ms/step" $GRAFT_REPO_ROOT/gpurun_out/$tag.log
