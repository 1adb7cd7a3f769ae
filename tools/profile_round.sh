#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r01'): tests, bench, rocprof kernel
# stats of the same bench command, PMC passes (separate runs, no tracing domains mixed in).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_sq -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_sq.err
python tools/conv_bench.py > $O/conv_tiles.txt 2>&1
python tools/layers.py > $O/layers.txt 2>&1
python tools/op_bench.py > $O/op_bench.json 2>&1
cat $O/pytest_gpu.txt; cat $O/bench.json; cat $O/bench_train.json
