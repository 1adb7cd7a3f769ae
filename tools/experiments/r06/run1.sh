set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -k "tile_shapes_agree or eight_phase or 256x128 or stream_k" > gpurun_out/r06/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t1.log
tail -5 gpurun_out/r06/t1.log
timeout 600 python tools/conv_bench_bf16.py 0an,0a,8842,8842a,8844a > gpurun_out/r06/bench1.log 2>&1
cat gpurun_out/r06/bench1.log
