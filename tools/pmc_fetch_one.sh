#!/bin/bash
# Run ON THE GPU BOX: fabric-side bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and L2 hit rate of ONE conv layer.
#   bash tools/pmc_fetch_one.sh <out name> f32|bf16     (env CONV_SHAPE = N,H,W,Cin,Cout,k for the 16-bit script)
# under rocprofv3 the profiler's preloaded library initialises HIP before Python runs: the queue count must be
# in the environment already (bench.py / the tools only `setdefault` it for unprofiled runs)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; dt=${2:-f32}
O=gpurun_out/$name; mkdir -p $O
script=tools/conv_one.py; [ "$dt" != "f32" ] && script=tools/conv_one_bf16.py
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -f csv -d $O/p$i -o one -- python3 $script > /dev/null 2> $O/p$i.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for f in glob.glob('$O/p*/one_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'conv_pp' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for f in glob.glob('$O/p1/one_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'conv_pp' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
v = {k: agg[k] / n[k] for k in agg}
print('$dt ${CONV_SHAPE:-default}: kernel us', [round(d, 1) for d in dur][:4])
print(f"  fetch = {2 * v.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB (FETCH_SIZE KiB x 2), write = {v.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB, "
      f"L2 hit rate = {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):.3f}")
PY
rm -rf $O/p1 $O/p2 $O/p3
