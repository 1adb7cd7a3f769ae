"""Where the device idles inside a train step: reads a rocprofv3 kernel trace CSV (`--kernel-trace -f csv`), takes the
last full step (between two `sgd_update_kernel` groups), and prints the union of busy time over all streams, the idle
gaps above a threshold with the kernels around them, and the time during which only small-grid kernels ran.
    python tools/experiments/gap_report.py <dir>/step_kernel_trace.csv [gap_us=5]"""
import csv, sys, re

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
ks = []
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name).split('(')[0][:60]
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, grid // max(wg, 1), r.get('Queue_Id', '')))
ks.sort()
marks = [i for i, k in enumerate(ks) if k[2].startswith('sgd_update_kernel')]
# groups of consecutive sgd launches = step ends
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or ks[marks[i + 1]][0] - ks[marks[i]][1] > 2e6]
if len(ends) < 2:
    sys.exit('need two optimizer steps in the trace')
a, b = ends[-2] + 1, ends[-1] + 1
step = [k for k in ks[a:b]]
t0, t1 = step[0][0], max(k[1] for k in step)
print(f'step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches, kernel time sum {sum(k[1] - k[0] for k in step) / 1e6:.3f} ms')
busy, cur_s, cur_e = 0, None, None
gaps = []
last_name = None
for s, e, n, wgs, q in step:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, n
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n, (cur_e - t0) / 1e6))
        cur_s, cur_e = s, e
        last_name = n
    else:
        if e > cur_e:
            cur_e, last_name = e, n
busy += cur_e - cur_s
print(f'busy (any stream) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps')
big = sorted([g for g in gaps if g[0] >= thr * 1e3], reverse=True)
print(f'gaps >= {thr} us: {len(big)}, total {sum(g[0] for g in big) / 1e6:.3f} ms')
for g in big[:40]:
    print(f'  {g[0] / 1e3:8.1f} us at {g[3]:7.3f} ms  after {g[1]}  before {g[2]}')
# time when every running kernel has fewer than 256 workgroups (the device is mostly empty)
ev = []
for s, e, n, wgs, q in step:
    ev.append((s, 1, wgs))
    ev.append((e, -1, wgs))
ev.sort()
act, small_t, prev = [], 0, None
import collections
cnt = collections.Counter()
for t, d, wgs in ev:
    if prev is not None and cnt and sum(w * c for w, c in cnt.items()) < 256:
        small_t += t - prev
    if d == 1:
        cnt[wgs] += 1
    else:
        cnt[wgs] -= 1
        if cnt[wgs] == 0:
            del cnt[wgs]
    prev = t
print(f'time with fewer than 256 workgroups in flight (all streams): {small_t / 1e6:.3f} ms')
small = collections.defaultdict(lambda: [0, 0])
for s, e, n, wgs, q in step:
    if wgs < 256:
        small[n][0] += 1
        small[n][1] += e - s
for n, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {t / 1e3:8.1f} us x{c:3d}  {n}')
