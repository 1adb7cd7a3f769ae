#!/bin/bash
# run on the GPU box: tests, bench, rocprof kernel stats, PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r01/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r01/stats -o step -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01/bench_under_rocprof.json 2> gpurun_out/r01/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d gpurun_out/r01/pmc_fetch -o step -- python3 tools/prof_step.py > /dev/null 2> gpurun_out/r01/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d gpurun_out/r01/pmc_write -o step -- python3 tools/prof_step.py > /dev/null 2> gpurun_out/r01/pmc_write.err
python tools/conv_bench.py > gpurun_out/r01/conv_tiles.txt 2>&1
python tools/layers.py > gpurun_out/r01/layers.txt 2>&1
ls -R gpurun_out/r01 | head -40
cat gpurun_out/r01/pytest_gpu.txt; cat gpurun_out/r01/bench.json
