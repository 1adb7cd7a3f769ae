"""kernel statistics from a rocprofv3 rocpd database (the default output when `-f csv` was not given):
prints launches per step and the top kernels, optionally writes a *_kernel_stats.csv like `--stats -f csv`"""
import csv
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name = 'display_name' if 'display_name' in cols else 'kernel_name'
    q = f'select s.{name}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) ' \
        f'from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name} order by 3 desc'
    return list(c.execute(q))


if __name__ == '__main__':
    rows = stats(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tot = sum(r[2] for r in rows)
    calls = sum(r[1] for r in rows)
    print(f'{calls} launches, {tot / 1e6:.2f} ms kernel time; per step ({steps}): {calls / steps:.0f} launches, '
          f'{tot / 1e6 / steps:.2f} ms')
    foreign = ('at::native', 'rocclr', 'rocprim', 'hipcub', 'ncclDevKernel', 'rccl')
    own = sum(r[2] for r in rows if not any(f in r[0] for f in foreign))
    print(f'own kernels: {100.0 * own / tot:.1f} % of kernel time')
    for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
        print(f'{r[1]:7d} {r[2] / 1e6:9.3f} ms {100.0 * r[2] / tot:6.2f}%  {r[0][:130]}')
    if len(sys.argv) > 3 and sys.argv[3] != '-':
        with open(sys.argv[3], 'w', newline='') as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for r in rows:
                w.writerow([r[0], r[1], r[2], r[2] / r[1], f'{100.0 * r[2] / tot:.2f}', r[3], r[4]])
