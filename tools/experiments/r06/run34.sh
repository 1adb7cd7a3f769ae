# refresh of the judged summaries that the late kernel changes move (op bench, bench line, kernel statistics of both legs)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python tools/op_bench.py > $O/op_bench.json 2> $O/op_bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf -o step -- python3 bench.py --mode inference --no-cpu-baseline --no-bs1 > $O/bench_inf_under_rocprof.json 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train_bf16 -o step -- python3 bench.py --mode train --train-dtype bf16 --steps 10 --warmup 3 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>> $O/rocprof_stats.err
python tools/train_layers.py > $O/train_layers_bf16.txt 2>&1
cut -c1-300 $O/bench.json
