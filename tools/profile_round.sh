#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02'): tests, bench, rocprof kernel stats of the same
# bench commands, PMC passes (separate runs, --kernel-trace only: no tracing domains mixed in).
# under rocprofv3 the profiler's preloaded library initialises HIP before Python runs: the queue count must be
# in the environment already (bench.py / the tools only `setdefault` it for unprofiled runs)
export GPU_MAX_HW_QUEUES=8
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --dtype bf16 --train-dtype f32 --no-cpu-baseline > $O/bench_bf16_trainf32.json 2>> $O/bench.err
python bench.py --dtype f16 --train-dtype f16 --no-cpu-baseline > $O/bench_f16.json 2>> $O/bench.err
# kernel statistics of the SAME default bench command (inference headline + train step)
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o step -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
# per-pass kernel statistics of the batch-8 headline workload alone (no batch-1 latency loop: `roofline` of the bench line
# follows from this file), and of the batch-1 loop separately
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf -o step -- python3 bench.py --mode inference --no-cpu-baseline --no-bs1 > $O/bench_inf_under_rocprof.json 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf_bs1 -o step -- python3 bench.py --mode inference --batch 1 --no-cpu-baseline --no-bs1 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train_bf16 -o step -- python3 bench.py --mode train --train-dtype bf16 --steps 10 --warmup 3 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_train_f32 -o step -- python3 bench.py --mode train --train-dtype f32 --steps 10 --warmup 3 > /dev/null 2>> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats -f csv -d $O/stats_inf_bf16 -o step -- python3 bench.py --mode inference --dtype bf16 --no-cpu-baseline --no-bs1 > /dev/null 2>> $O/rocprof_stats.err
# PMC: conv traffic / MFMA utilisation (inference pass), op-level kernels
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_sq -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_sq.err
BRCNN_DTYPE=bf16 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_sq_bf16 -o step -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_sq_bf16.err
# bf16 train step: HBM traffic and MFMA utilisation of the conv / weight-gradient kernels
BRCNN_DTYPE=bf16 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_train_fetch -o step -- python3 tools/prof_train.py > /dev/null 2> $O/pmc_train_fetch.err
BRCNN_DTYPE=bf16 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_train_write -o step -- python3 tools/prof_train.py > /dev/null 2> $O/pmc_train_write.err
BRCNN_DTYPE=bf16 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $O/pmc_train_sq -o step -- python3 tools/prof_train.py > /dev/null 2> $O/pmc_train_sq.err
# per-layer PMC of the 3x3 layers of stages 2 / 3 / 4 and the tower (heuristic tile)
( CONV_SHAPE=8,100,168,128,128,3 bash tools/pmc_conv_one.sh $R/pmc_l_s2 0; CONV_SHAPE=8,50,84,256,256,3 bash tools/pmc_conv_one.sh $R/pmc_l_s3 0; CONV_SHAPE=8,25,42,512,512,3 bash tools/pmc_conv_one.sh $R/pmc_l_s4 0; CONV_SHAPE=8,140,160,256,256,3 bash tools/pmc_conv_one.sh $R/pmc_l_tower 0 ) > $O/conv_pmc_layers_bf16.txt 2>&1
python tools/op_bench.py > $O/op_bench.json 2> $O/op_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/op_pmc_fetch -o step -- python3 tools/op_bench.py > /dev/null 2> $O/op_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/op_pmc_write -o step -- python3 tools/op_bench.py > /dev/null 2> $O/op_pmc_write.err
python tools/wgrad_pp_bench.py 75 > $O/wgrad_pp_bench.txt 2>&1
python tools/experiments/roi_variants.py > $O/roi_variants.txt 2>&1
python tools/layers.py > $O/layers.txt 2>&1
BRCNN_DTYPE=bf16 python tools/layers.py > $O/layers_bf16.txt 2>&1
python tools/train_layers.py > $O/train_layers_bf16.txt 2>&1
BRCNN_DTYPE=f32 python tools/train_layers.py > $O/train_layers_f32.txt 2>&1
python tools/experiments/r06/stem_bench.py > $O/stem_bench.txt 2>&1
python tools/bench_recipes.py > $O/recipes.txt 2>&1
cp gpurun_out/recipes.json $O/recipes.json 2>/dev/null
cat $O/pytest_gpu.txt; cut -c1-400 $O/bench.json; ls $O/stats
