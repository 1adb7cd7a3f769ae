"""copy the judged summaries of `tools/profile_round.sh <round>` from gpurun_out/<round>/ into profiles/"""
import os, shutil, subprocess, sys
r = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src, dst = f'gpurun_out/{r}', 'profiles'
pairs = [('bench_f16.json', 'bench_f16.json'), ('bench_bf16_trainf32.json', 'bench_bf16_trainf32.json'),
         ('bench_under_rocprof.json', 'bench_under_rocprof.json'),
         ('stats_inf/step_kernel_stats.csv', 'inference_kernel_stats.csv'),
         ('stats_inf_bs1/step_kernel_stats.csv', 'inference_bs1_kernel_stats.csv'),
         ('bench_inf_under_rocprof.json', 'bench_inf_under_rocprof.json'),
         ('stats_inf_bf16/step_kernel_stats.csv', 'inference_bf16_kernel_stats.csv'),
         ('stats_train_f32/step_kernel_stats.csv', 'train_f32_kernel_stats.csv'),
         ('train_layers_bf16.txt', 'train_layers_bf16.txt'), ('train_layers_f32.txt', 'train_layers_f32.txt'),
         ('recipes.txt', 'recipes.txt'),
         ('bench.json', 'bench.json'), ('bench_bf16.json', 'bench_bf16.json'), ('bench_train.json', 'bench_train.json'),
         ('bench_train_bf16.json', 'bench_train_bf16.json'), ('stats/step_kernel_stats.csv', 'bench_kernel_stats.csv'),
         ('stats_bf16/step_kernel_stats.csv', 'bench_bf16_kernel_stats.csv'),
         ('stats_train/step_kernel_stats.csv', 'train_kernel_stats.csv'),
         ('stats_train_bf16/step_kernel_stats.csv', 'train_bf16_kernel_stats.csv'),
         ('conv_tiles.txt', 'conv_tiles.txt'), ('conv_tiles_bf16.txt', 'conv_tiles_bf16.txt'),
         ('layers.txt', 'conv_layers.txt'), ('layers_bf16.txt', 'conv_layers_bf16.txt'),
         ('op_bench.json', 'op_bench.json'), ('conv_pmc_layers_bf16.txt', 'conv_pmc_layers_bf16.txt'), ('recipes.json', 'recipes.json'), ('recipes_bf16.json', 'recipes_bf16.json'), ('pytest_gpu.txt', 'pytest_gpu.txt'),
         ('wgrad_pp_bench.txt', 'wgrad_pp_bench.txt'), ('stem_bench.txt', 'stem_bench.txt'), ('roi_variants.txt', 'roi_variants.txt')]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f'{r}_{b}'))
        print('copied', a)
    else:
        print('MISSING', a)
subprocess.check_call([sys.executable, 'tools/summarize_pmc.py', r])
if os.path.isdir(os.path.join(src, 'pmc_train_fetch')):
    subprocess.check_call([sys.executable, 'tools/summarize_train_pmc.py', r, 'bf16'])
if os.path.isdir(os.path.join(src, 'op_pmc_fetch')):
    subprocess.check_call([sys.executable, 'tools/summarize_op_pmc.py', r])
