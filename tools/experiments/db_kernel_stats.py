"""print a rocprofv3 kernel-stats CSV per train / inference step: python tools/experiments/db_kernel_stats.py <csv> <steps> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
own = 0.0
for r in rows:
    n = r['Name']
    if not (n.startswith('void at::') or n.startswith('at::') or '__amd_rocclr' in n or 'rocprim' in n or 'hipcub' in n.lower()):
        own += float(r['TotalDurationNs'])
print(f'kernel time per step {tot / steps / 1e6:.3f} ms, launches per step {sum(int(r["Calls"]) for r in rows) / steps:.0f}, own kernels {100 * own / tot:.1f} %')
for r in rows[:top]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.2f}%")
