"""host-side profile of bench.py's OWN train step (FusedSGD, early RPN backward, arena): cProfile over the timed steps,
plus the per-step statistics of the bench line.  Where the Python / launch overhead of the second half of the step
(everything behind the sampler's host synchronisation) goes decides whether the step is device- or launch-bound."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['bench.py', '--mode', 'train', '--steps', '10', '--warmup', '5', '--no-cpu-baseline'] + sys.argv[1:]
import bench
_orig = bench.timed


def timed(step, steps, warmup, world, device, after_warmup=None):
    for _ in range(warmup):          # (the profile covers the timed steps only)
        step()
    warmup = 0
    pr = cProfile.Profile()
    pr.enable()
    out = _orig(step, steps, warmup, world, device, after_warmup=after_warmup)
    pr.disable()
    for key, n in (('tottime', 40), ('cumulative', 70)):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
        print(s.getvalue()[:14000], file=sys.stderr)
    return out


bench.timed = timed
bench.main()
