"""aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid): python tools/experiments/trace_by_grid.py <kernel_trace.csv> [substring]"""
import collections, csv, sys
rows = collections.defaultdict(list)
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if sub and sub not in n:
        continue
    short = n.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    g = (r.get('Grid_Size_X') or r.get('Grid_Size') or '?', r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '?', r.get('Grid_Size_Y', '1'))
    rows[(short, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
for (k, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f'{k:72s} grid {g[0]:>8s}x{g[2]:>3s} wg {g[1]:>5s} n={len(v):4d} med {v2[len(v2)//2]:8.1f} us min {v2[0]:8.1f} tot {sum(v)/1000:8.2f} ms')
