set -x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05prof_bench -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-operating-points > $GRAFT_REPO_ROOT/gpurun_out/r05_bench_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_bench_rocprof.err
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/gpurun_out/r05prof_bench/r_results.db $GRAFT_REPO_ROOT/gpurun_out/r05_bench_rocprof.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05prof_bench
cd $GRAFT_REPO_ROOT
bash scripts/pmc_gemm.sh r05_pmc_ring_gemm 24570 0 > gpurun_out/r05_pmc_ring_gemm.txt 2>&1
bash scripts/pmc_fetch.sh r05_pmc_ring_fetch scripts/bench_gemm.py 24570 0 > /dev/null 2>&1
bash scripts/pmc_attn.sh r05_pmc_attn_decode 448 830 > /dev/null 2>&1

rm -rf gpurun_out/r05_pmc_ring_gemm gpurun_out/*.FETCH_SIZE gpurun_out/*.WRITE_SIZE
head -c 600 gpurun_out/r05_bench_rocprof.json; tail -5 gpurun_out/r05_pmc_ring_gemm.txt; cat gpurun_out/r05_pmc_attn_decode.txt
