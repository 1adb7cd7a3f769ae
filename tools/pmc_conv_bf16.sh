# under rocprofv3 the profiler's preloaded library initialises HIP before Python runs: the queue count must be
# in the environment already (bench.py / the tools only `setdefault` it for unprofiled runs)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcbf; mkdir -p $O
i=0
for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -f csv -d $O/p$i -o one -- python3 tools/conv_one_bf16.py > /dev/null 2> $O/p$i.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmcbf/p*/one_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm_bf16' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(agg): print(k, agg[k] / n[k], n[k])
PY
tail -3 gpurun_out/pmcbf/p3.err
