# periodic host stalls of the train step (+4-8 ms every ~100 ms on some boxes): CPU-quota throttling by spinning worker
# threads?  OMP_NUM_THREADS=1 against the default, interleaved; prints mean / median and the number of steps > 1.1 x median
for i in 1 2 3; do
for v in "BRCNN_NOP=1" "OMP_NUM_THREADS=1 MKL_NUM_THREADS=1" "OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0"; do
echo "$v"
env $v BRCNN_BENCH_DUMP_STEPS=1 timeout 200 python bench.py --mode train --steps 40 --warmup 8 2>&1 >/dev/null | grep "bench: steps" | python3 -c "
import sys
for l in sys.stdin:
    v=[float(x.split('/')[0]) for x in l.split()[2:]]
    h=[float(x.split('/')[1]) for x in l.split()[2:]]
    m=sorted(v)[len(v)//2]
    print('   mean %.2f median %.2f  slow steps %d of %d  host max %.1f' % (sum(v)/len(v), m, sum(1 for x in v if x>1.1*m), len(v), max(h)))
"
done; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
