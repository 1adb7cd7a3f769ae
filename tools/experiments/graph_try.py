"""Try capturing the whole device-resident inference pass in a HIP graph; compare with eager."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
m, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', torch.device('cuda', 0))
m.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'f32'))
img, metas = bench.synthetic_batch(8, 'cuda')
def run():
    with torch.no_grad():
        return m.simple_test_device(img, metas, rescale=True)
for _ in range(3):
    ref = run()
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        o = fn(); o[2].cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print('eager ms', timeit(run))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        run()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = run()
except Exception as e:
    import traceback; traceback.print_exc()
    print('CAPTURE FAILED:', repr(e)[:300])
    sys.exit(0)
def replay():
    g.replay()
    return out
o = replay()
torch.cuda.synchronize()
print('match', [bool(torch.equal(a, b)) for a, b in zip(o, ref)])
print('graph ms', timeit(replay))
