"""Builds libbrcnn_hip.so (every HIP kernel of the hot path + the C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(`boosting-r-cnn_amd/lib/libbrcnn_hip.so`): it is git-ignored but travels to the GPU box with
the repository snapshot.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libbrcnn_hip.so')
ARCH = 'gfx950'
SOURCES = ['roi_align.hip', 'nms.hip', 'soft_nms.hip', 'focal_loss.hip', 'conv_igemm.hip', 'conv_igemm_bf16.hip', 'conv_wgrad.hip',
           'misc.hip', 'rpn.hip']
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-fvisibility=hidden', '-Wno-unused-result',
         # hardware fp32 atomic add (global_atomic_add_f32) instead of a CAS loop for the gradient scatters
         '-munsafe-fp-atomics']


# per-source additions.  -fno-slp-vectorize: in these translation units every packed-fp32 instruction came from the SLP
# vectoriser, some of them in the cross-swizzled op_sel form that tools/check_isa.py refuses; the kernels
# are latency- or HBM-bound, the scalar form costs nothing measurable
EXTRA_FLAGS = {'roi_align.hip': ['-fno-slp-vectorize'], 'focal_loss.hip': ['-fno-slp-vectorize'],
               'train_loss.hip': ['-fno-slp-vectorize'], 'deform.hip': ['-fno-slp-vectorize']}


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: cannot build libbrcnn_hip.so')


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + \
        sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') and f not in SOURCES)


def _stamp():
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + ['../../include/brcnn_hip.h']:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _obj_stamp(src):
    """what an object file depends on: its source, every header of csrc/ and the public header, the flags"""
    h = hashlib.sha1()
    for f in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + ['../../include/brcnn_hip.h']:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), 'rb').read())
    h.update(' '.join(FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _compile(src, force=False):
    obj = os.path.join(LIBDIR, 'obj', src.replace('.hip', '.o'))
    stamp_file, stamp = obj + '.stamp', _obj_stamp(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj                      # unchanged since it was compiled (sources are compiled one by one)
    cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stderr[-4000:]}')
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    return obj


def build_library(force=False, verbose=False):
    os.makedirs(os.path.join(LIBDIR, 'obj'), exist_ok=True)
    stamp_file = os.path.join(LIBDIR, 'stamp')
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and \
            open(stamp_file).read() == stamp:
        return LIB
    srcs = _sources()
    objdir = os.path.join(LIBDIR, 'obj')
    for f in os.listdir(objdir):            # objects of sources that no longer exist
        if f.endswith('.o') and f.replace('.o', '.hip') not in srcs:
            os.remove(os.path.join(objdir, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda src: _compile(src, force), srcs))
    # ISA gate: no cross-swizzled packed-fp32 instruction in any code object (check_isa.py, in this package; skipped with
    # a warning when the ROCm installation in use has no llvm-objdump)
    import importlib.util
    spec = importlib.util.spec_from_file_location('brcnn_check_isa', os.path.join(HERE, 'check_isa.py'))
    check_isa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(check_isa)
    check_isa.check_objects(objs, verbose=verbose, hipcc=hipcc())
    cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
