#!/bin/bash
# Run ON THE GPU BOX: PMC passes (separate runs, no tracing domains) over the largest fp32 conv layer
# under rocprofv3 the profiler's preloaded library initialises HIP before Python runs: the queue count must be
# in the environment already (bench.py / the tools only `setdefault` it for unprofiled runs)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcf32; mkdir -p $O
i=0
for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -f csv -d $O/p$i -o one -- python3 tools/conv_one.py > /dev/null 2> $O/p$i.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for f in glob.glob('gpurun_out/pmcf32/p*/one_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm_f32' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for f in glob.glob('gpurun_out/pmcf32/p*/one_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm_f32' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(agg): print(k, agg[k] / n[k], n[k])
d = sorted(dur)[len(dur) // 2]
print('median kernel ns', d)
if 'SQ_BUSY_CYCLES' in agg:
    busy = agg['SQ_BUSY_CYCLES'] / n['SQ_BUSY_CYCLES'] / 32.0       # summed over 32 shader engines
    mf = agg['SQ_VALU_MFMA_BUSY_CYCLES'] / n['SQ_VALU_MFMA_BUSY_CYCLES']
    print('mfma util', mf / (1024 * busy), 'effective GHz (SQ_BUSY per SE / wall)', busy / d)
if 'GRBM_GUI_ACTIVE' in agg:
    print('effective GHz (GRBM_GUI_ACTIVE / wall)', agg['GRBM_GUI_ACTIVE'] / n['GRBM_GUI_ACTIVE'] / d)
PY
