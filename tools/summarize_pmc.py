"""Summarise the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs of
tools/prof_step.py = 4 inference passes) into profiles/r01_conv_traffic.json.
Units/corrections per MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads -> doubled."""
import collections, csv, json, re, sys


def short_name(full):
    """`void (anonymous namespace)::gn_apply_rows_kernel<float>((anonymous namespace)::Seg, ...)` -> `gn_apply_rows_kernel`;
    `void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, ...>(...)` ->
    `at::native::vectorized_elementwise_kernel<FillFunctor>`: the function's own name (the identifier in front of its
    template / argument list, namespaces of our own kernels dropped) plus, for torch's generic element-wise kernels, the
    functor that tells them apart"""
    n = full.strip()
    n = re.sub(r'^(void|int|float)\s+', '', n)
    n = n.replace('(anonymous namespace)::', '')
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):          # the first '<' or '(' at nesting depth 0 ends the function name
        if ch in '<(' and depth == 0:
            cut = i
            break
    base = n[:cut].strip()
    if base.startswith('at::native::') and ('elementwise' in base or 'reduce_kernel' in base):
        m = re.search(r'(\w+Functor\w*|\w+_kernel_cuda\w*|\w+Ops?\b)', n[cut:])
        if m:
            base += f'<{m.group(1)}>'
    return base or n[:60]


rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for name in ('fetch', 'write'):
    for r in csv.DictReader(open(f'gpurun_out/{rnd}/pmc_{name}/step_counter_collection.csv')):
        # every kernel under its own name (conv_pp_f32_kernel, conv_igemm_f32_dma_kernel, ... separately), and every
        # conv / FC launch of the pass once more under the aggregate key the bench line's `roofline.traffic` reads
        keys = [short_name(r['Kernel_Name'])]
        # (round 6: the stem + max-pool launch and the fused stage-1 block tails are conv launches of the pass too -- the bench
        # line's roofline counts their products)
        if any(k in r['Kernel_Name'] for k in ('conv_igemm', 'conv_pp_', 'conv1x1_stream', 'stem_pool_kernel', 'bottleneck_tail')):
            keys.append('conv_stack_all_launches')
        for k in keys:
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            if name == 'fetch':
                cnt[k] += 1
out = {}
for k, v in tot.items():
    n = max(cnt[k], 1)
    fetch_b = 2.0 * v.get('FETCH_SIZE', 0.0) * 1024
    write_b = v.get('WRITE_SIZE', 0.0) * 1024
    out[k] = dict(launches=n, fetch_bytes_per_launch=fetch_b / n, write_bytes_per_launch=write_b / n,
                  hbm_bytes_per_launch=(fetch_b + write_b) / n)
top = dict(sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:14])
json.dump(dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/prof_step.py, '
                      '4 inference passes of batch 8; FETCH_SIZE doubled per MI355X_MICROARCH.md',
               kernels=top), open(f'profiles/{rnd}_conv_traffic.json', 'w'), indent=1)
print(json.dumps(top['conv_stack_all_launches'], indent=1))

# ---- MFMA utilisation of the conv kernels from the SQ pass (if present) ---------------------
try:
    agg = collections.defaultdict(float)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n_conv = 0
    for r in csv.DictReader(open(f'gpurun_out/{rnd}/pmc_sq/step_counter_collection.csv')):
        for key in ('conv_igemm', 'conv_pp_', 'stem_pool_kernel', 'bottleneck_tail_f32_kernel'):
            if key in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value'])
                per[key][r['Counter_Name']] += float(r['Counter_Value'])
    # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs share the MFMA cycles
    clk_cycles = agg['SQ_BUSY_CYCLES'] / 32.0
    util = agg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * clk_cycles)
    sq = dict(mfma_busy_cycles=agg['SQ_VALU_MFMA_BUSY_CYCLES'], sq_busy_cycles_per_se=clk_cycles,
              mfma_util=util, wait_any_frac=agg['SQ_WAIT_ANY'] / agg['SQ_WAVE_CYCLES'],
              wait_inst_any_frac=agg['SQ_WAIT_INST_ANY'] / agg['SQ_WAVE_CYCLES'],
              by_kernel={k.rstrip('_'): dict(mfma_util=v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * v['SQ_BUSY_CYCLES'] / 32.0),
                                             wait_any_frac=v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'])
                         for k, v in per.items() if v.get('SQ_BUSY_CYCLES')},
              note='all conv launches (conv_igemm_f32*, conv_pp_f32) of 4 inference passes; MfmaUtil = '
                   'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES/32)')
    json.dump(sq, open(f'profiles/{rnd}_conv_mfma_util.json', 'w'), indent=1)
    print('mfma_util', util)
except FileNotFoundError:
    pass
