"""per-kernel VALU / MFMA occupancy of a rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES pass:
python tools/experiments/valu_by_kernel.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [steps]"""
import csv, glob, sys, collections, re
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''))[:90]
        cnt[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_BUSY_CYCLES': calls[k] += 1
rows = []
for k, v in cnt.items():
    clk = v['SQ_BUSY_CYCLES'] / 32.0
    if clk <= 0: continue
    simd = 1024.0 * clk
    rows.append((clk, k, 4 * v['SQ_INSTS_VALU'] / simd, v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd, 4 * v['SQ_WAVE_CYCLES'] / simd, calls[k]))
tot = sum(r[0] for r in rows)
print('kernel | share of busy cycles | launches/step | VALU busy | MFMA busy | waves/SIMD')
for clk, k, valu, mfma, occ, n in sorted(rows, reverse=True)[:40]:
    print(f'{k:90s} {100 * clk / tot:5.1f}% {n / steps:6.1f} {valu:5.2f} {mfma:5.2f} {occ:5.2f}')
