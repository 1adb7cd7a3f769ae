export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --dtype bf16 --train-dtype f32 --no-cpu-baseline > $O/bench_bf16_trainf32.json 2>> $O/bench.err
python bench.py --dtype f16 --train-dtype f16 --no-cpu-baseline > $O/bench_f16.json 2>> $O/bench.err
python tools/experiments/r06/tail_bench.py > $O/tail_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf -o step -- python3 bench.py --mode inference --no-cpu-baseline --no-bs1 > $O/bench_inf_under_rocprof.json 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train_bf16 -o step -- python3 bench.py --mode train --train-dtype bf16 --steps 10 --warmup 3 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf_bf16 -o step -- python3 bench.py --mode inference --dtype bf16 --no-cpu-baseline --no-bs1 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>> $O/rocprof_stats.err
python tools/layers.py > $O/layers.txt 2>&1
BRCNN_DTYPE=bf16 python tools/layers.py > $O/layers_bf16.txt 2>&1
python tools/train_layers.py > $O/train_layers_bf16.txt 2>&1
cat $O/pytest_gpu.txt; cut -c1-200 $O/bench.json
