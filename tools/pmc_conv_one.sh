# PMC passes of ONE 16-bit conv layer (tools/conv_one_bf16.py); usage: bash tools/pmc_conv_one.sh <out name> [tile ...]
# env CONV_SHAPE = N,H,W,Cin,Cout,k; CONV_DT = bf16 / f16 / f32; CONV_RES=1: with a residual operand.  Counters in separate passes (MI355X_MICROARCH.md: slots per pass).
# under rocprofv3 the profiler's preloaded library initialises HIP before Python runs: the queue count must be
# in the environment already (bench.py / the tools only `setdefault` it for unprofiled runs)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
for tile in "$@"; do
  export BF16_TILE=$tile
  O=gpurun_out/$name/t$tile; mkdir -p $O
  i=0
  for c in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace -f csv -d $O/p$i -o one -- python3 tools/conv_one_bf16.py > /dev/null 2> $O/p$i.err
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
dur = []
for f in glob.glob('$O/p*/one_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name'] and ('igemm' in r['Kernel_Name'] or 'conv_pp' in r['Kernel_Name'] or 'stream' in r['Kernel_Name']):
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for f in glob.glob('$O/p1/one_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name'] and ('igemm' in r['Kernel_Name'] or 'conv_pp' in r['Kernel_Name'] or 'stream' in r['Kernel_Name']):
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('tile $tile shape ${CONV_SHAPE:-default}: kernel us (profiled)', [round(d, 1) for d in dur])
v = {k: agg[k] / n[k] for k in agg}
for k in sorted(v): print(f'  {k:32s} {v[k]:16.0f}')
if 'SQ_BUSY_CYCLES' in v:
    clk = v['SQ_BUSY_CYCLES'] / 32.0       # summed over 32 shader engines
    print(f"  MfmaUtil = MFMA_BUSY / (1024 SIMDs x busy cycles) = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * clk):.3f}")
    print(f"  wait_any / wave_cycles = {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, wait_inst_any = {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, active_inst_any = {v.get('SQ_ACTIVE_INST_ANY', 0) / v['SQ_WAVE_CYCLES']:.3f}")
    if 'SQ_INSTS_VALU' in v: print(f"  VALU busy = 4 x SQ_INSTS_VALU / (1024 SIMDs x busy cycles) = {4 * v['SQ_INSTS_VALU'] / (1024.0 * clk):.3f}; waves per SIMD = 4 x wave quad-cycles / SIMD cycles = {4 * v['SQ_WAVE_CYCLES'] / (1024.0 * clk):.2f}")
    if dur: print(f"  effective clock = busy cycles / duration = {clk / (sum(dur) / len(dur)) / 1e3:.2f} GHz")
if 'TCC_HIT_sum' in v: print(f"  L2 hit rate = {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
if 'FETCH_SIZE' in v: print(f"  HBM fetch = {2 * v['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (FETCH_SIZE KiB x 2, gfx950), write = {v.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
PY
done
