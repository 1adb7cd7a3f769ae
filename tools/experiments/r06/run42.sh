export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf -o step -- python3 bench.py --mode inference --no-cpu-baseline --no-bs1 > $O/bench_inf_under_rocprof.json 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf_bs1 -o step -- python3 bench.py --mode inference --batch 1 --no-cpu-baseline --no-bs1 > /dev/null 2>> $O/rocprof_stats.err
python tools/layers.py > $O/layers.txt 2>&1
python tools/experiments/r06/tail_bench.py > $O/tail_bench.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_sq -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_sq.err
cut -c1-260 $O/bench.json; cat $O/tail_bench.txt | tail -2
