# r06 (512 clips per step as in r05): kernel summary of `python bench.py` (no flags but the two that skip the CPU sample and the
# operating points) and the decode attention's PMC passes at the benchmark shape (512 clips, mean context 830)
set -x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06prof_bench -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-operating-points > $GRAFT_REPO_ROOT/gpurun_out/r06_bench_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_bench_rocprof.err
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/gpurun_out/r06prof_bench/r_results.db $GRAFT_REPO_ROOT/gpurun_out/r06_bench_rocprof.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r06prof_bench
cd $GRAFT_REPO_ROOT
bash scripts/pmc_attn.sh r06_pmc_attn_decode 512 830 > /dev/null 2>&1
rm -rf gpurun_out/*.FETCH_SIZE gpurun_out/*.WRITE_SIZE
head -c 400 gpurun_out/r06_bench_rocprof.json; head -30 gpurun_out/r06_bench_rocprof.txt; cat gpurun_out/r06_pmc_attn_decode.txt
