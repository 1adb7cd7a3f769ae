export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/trace_a; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O -o step -- python3 bench.py --mode train --train-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
ls $O; tail -c 600 $O/bench.json
