"""PMC passes of the bf16 train step (tools/profile_round.sh: pmc_train_fetch / pmc_train_write / pmc_train_sq, separate
rocprofv3 runs of tools/prof_train.py) -> profiles/<round>_conv_traffic_train_bf16.json and
profiles/<round>_conv_mfma_util_bf16.json.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the
bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md).  SQ_BUSY_CYCLES is summed over the 32 shader engines,
the MFMA busy cycles over the 1024 SIMDs."""
import collections, csv, json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
dt = sys.argv[2] if len(sys.argv) > 2 else 'bf16'


def family(name):
    for k in ('conv_wgrad_pp_bf16_kernel', 'wgrad_pp_reduce_kernel', 'conv_pp_bf16_kernel', 'conv_igemm_bf16_dma_kernel',
              'conv_wgrad_bf16_kernel', 'wgrad_reduce_kernel',
              'conv_igemm_f32', 'conv_wgrad_f32'):
        if k in name:
            return k
    return None


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for name in ('fetch', 'write', 'sq'):
    try:
        rows = csv.DictReader(open(f'gpurun_out/{rnd}/pmc_train_{name}/step_counter_collection.csv'))
    except FileNotFoundError:
        continue
    seen = set()
    for r in rows:
        f = family(r['Kernel_Name'])
        if f is None:
            continue
        tot[f][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'], f)
        if name == 'fetch' and key not in seen:
            seen.add(key)
            cnt[f] += 1
traffic, util = {}, {}
all_fetch = all_write = 0.0
all_n = 0
for f, v in tot.items():
    n = max(cnt[f], 1)
    fb, wb = 2.0 * v.get('FETCH_SIZE', 0.0) * 1024, v.get('WRITE_SIZE', 0.0) * 1024
    traffic[f] = dict(launches=cnt[f], fetch_bytes_per_launch=fb / n, write_bytes_per_launch=wb / n, hbm_bytes_per_launch=(fb + wb) / n)
    all_fetch += fb; all_write += wb; all_n += cnt[f]
    if v.get('SQ_BUSY_CYCLES'):
        clk = v['SQ_BUSY_CYCLES'] / 32.0
        util[f] = dict(mfma_util=v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * clk), wait_any_frac=v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'],
                       wait_inst_any_frac=v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'])
sq_all = collections.defaultdict(float)
for f, v in tot.items():
    for k in ('SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES'):
        sq_all[k] += v.get(k, 0.0)
src = ('rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE / SQ passes, --kernel-trace only) on tools/prof_train.py: 4 train '
       f'steps of boosting_rcnn_r50_pafpn_1x_coco.py, batch 8, {dt} conv stack; FETCH_SIZE doubled per MI355X_MICROARCH.md')
json.dump(dict(source=src, hbm_bytes_per_launch=(all_fetch + all_write) / max(all_n, 1), launches=all_n, kernels=traffic),
          open(f'profiles/{rnd}_conv_traffic_train_{dt}.json', 'w'), indent=1)
if sq_all['SQ_BUSY_CYCLES']:
    util['all conv / wgrad launches'] = dict(mfma_util=sq_all['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * sq_all['SQ_BUSY_CYCLES'] / 32.0))
json.dump(dict(source=src, kernels=util), open(f'profiles/{rnd}_conv_mfma_util_{dt}.json', 'w'), indent=1)
print(json.dumps(util, indent=1))
print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in traffic.items()}))
