export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace -f csv -d $O/op_pmc_l2 -o step -- python3 tools/op_bench.py > /dev/null 2> $O/op_pmc_l2.err
ls $O/op_pmc_l2 | head
