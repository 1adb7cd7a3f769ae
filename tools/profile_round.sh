#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r01'): tests, bench, rocprof kernel
# stats of the same bench command, PMC passes (separate runs, no tracing domains mixed in).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
python bench.py --mode train --steps 8 --warmup 3 > $O/bench_train.json 2>> $O/bench.err
python bench.py --mode train --dtype bf16 --steps 8 --warmup 3 > $O/bench_train_bf16.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train -o step -- python3 tools/prof_train.py > /dev/null 2> $O/rocprof_train.err
BRCNN_DTYPE=bf16 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train_bf16 -o step -- python3 tools/prof_train.py > /dev/null 2>> $O/rocprof_train.err
BRCNN_DTYPE=bf16 rocprofv3 --kernel-trace --stats -f csv -d $O/stats_bf16 -o step -- python3 tools/prof_step.py > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_sq -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_sq.err
python tools/conv_bench.py > $O/conv_tiles.txt 2>&1
python tools/conv_bench_bf16.py 11,21,22,81,82,0 > $O/conv_tiles_bf16.txt 2>&1
python tools/layers.py > $O/layers.txt 2>&1
BRCNN_DTYPE=bf16 python tools/layers.py > $O/layers_bf16.txt 2>&1
python tools/op_bench.py > $O/op_bench.json 2>$O/op_bench.err
python tools/bench_recipes.py > $O/recipes.txt 2>&1
cp gpurun_out/recipes.json $O/recipes.json
BRCNN_DTYPE=bf16 python tools/bench_recipes.py > $O/recipes_bf16.txt 2>&1
cp gpurun_out/recipes_bf16.json $O/recipes_bf16.json
cat $O/pytest_gpu.txt; cat $O/bench.json; cat $O/bench_bf16.json | cut -c1-250; cat $O/bench_train.json | cut -c1-250; cat $O/bench_train_bf16.json | cut -c1-250
