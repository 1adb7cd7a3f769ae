"""ISA gates of the build.  (1) ADVICE r04: no packed-fp32 VALU instruction may SWIZZLE a VGPR-pair operand; (2) round 6:
the listed streaming kernels keep their loads in flight in batches (`MIN_LOADS_IN_FLIGHT` below).

Round 4 traced an intermittent wrong dgamma of `bn_act_bwd_kernel` (3 of 400 repetitions, fp16, only with a
weight-gradient kernel co-resident) to `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` reading the result of a
`v_pk_mul_f32` written two instructions earlier (profiles/r04_notes.md): the listing's data flow was right, the result
depended on issue timing.  The kernel was reshaped until the compiler stopped emitting the pattern; nothing checked the
other translation units, several of which gained packed-fp32 epilogues in the same round.  This script disassembles the
gfx950 code object of every compiled source and fails the build on

    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32  with a VGPR-pair source whose op_sel bit is 1

i.e. the LOW result lane fed from the HIGH register of the pair -- the cross-swizzled form of the failure.  The default
is op_sel 0 / op_sel_hi 1 (both halves in their natural order).  The BROADCAST form (op_sel 0, op_sel_hi 0: both lanes
read the pair's low register; what a scalar scale / shift times two outputs compiles to) is counted and reported but
not refused: the conv read-outs hold ~5 000 of them, they never read the high register out of order, and the 200- to
400-repetition bit-reproducibility stress tests (tests/test_stress_gpu.py) run exactly those read-outs under a
co-resident load.  Where the compiler's SLP vectoriser was the only source of packed fp32 (roi_align, focal_loss,
train_loss, deform: latency- or HBM-bound kernels) the translation unit is built with -fno-slp-vectorize
(build.py EXTRA_FLAGS) and holds none at all.  `build.py` calls `check_objects` after compiling;
`python tools/check_isa.py [objects...]` (a shim over this module) runs it by hand and prints the per-object counts.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = None        # resolved by objdump(): next to the hipcc in use, $LLVM_OBJDUMP, PATH, /opt/rocm


def objdump(hipcc=None):
    """llvm-objdump of the ROCm installation whose hipcc builds the library (<prefix>/bin/hipcc ->
    <prefix>/lib/llvm/bin/llvm-objdump), else $LLVM_OBJDUMP, PATH, /opt/rocm; None when there is none"""
    global OBJDUMP
    if OBJDUMP and os.path.exists(OBJDUMP):
        return OBJDUMP
    cands = [os.environ.get('LLVM_OBJDUMP')]
    for h in (hipcc, os.environ.get('HIPCC'), shutil.which('hipcc')):
        if h:
            prefix = os.path.dirname(os.path.dirname(os.path.realpath(h)))
            cands += [os.path.join(prefix, 'lib', 'llvm', 'bin', 'llvm-objdump'), os.path.join(prefix, 'llvm', 'bin', 'llvm-objdump')]
    cands += [shutil.which('llvm-objdump'), '/opt/rocm/lib/llvm/bin/llvm-objdump']
    for c in cands:
        if c and os.path.exists(c):
            OBJDUMP = c
            return c
    return None
_PK = re.compile(r'^\s*(v_pk_(?:mul|add|fma)_f32)\s+(.*?)\s*//')
_SEL = re.compile(r'\b(op_sel|op_sel_hi):\[([01,]+)\]')
_SYM = re.compile(r'^[0-9a-f]+ <(.+)>:')


def device_code_object(obj, workdir):
    """extract the gfx950 code object of a hipcc-compiled host object into `workdir`; returns its path"""
    tmp = os.path.join(workdir, os.path.basename(obj))
    shutil.copy(obj, tmp)           # llvm-objdump --offloading writes the bundle entries next to its input
    subprocess.run([objdump(), '--offloading', tmp], check=True, capture_output=True)
    cands = [f for f in os.listdir(workdir) if f.startswith(os.path.basename(obj) + '.') and 'amdgcn' in f]
    if not cands:
        raise RuntimeError(f'{obj}: no amdgcn code object inside')
    return os.path.join(workdir, cands[0])


def swizzled_packed_fp32(code_object):
    """([(kernel symbol, instruction text)] of the refused instructions, count of packed-fp32 ones, count of
    broadcast-form ones)"""
    out = subprocess.run([objdump(), '-d', code_object], check=True, capture_output=True, text=True).stdout
    sym, bad, total, bcast = '?', [], 0, 0
    for line in out.splitlines():
        m = _SYM.match(line)
        if m:
            sym = m.group(1)
            continue
        m = _PK.match(line)
        if not m:
            continue
        total += 1
        text = m.group(2)
        # operands up to the first modifier; registers look like v[4:5] (a comma-free token)
        regs = []
        for o in re.split(r'\s+(?=op_sel|neg_|clamp)', text)[0].split(','):
            regs.append(o.strip())
        srcs = regs[1:]
        sel = {k: [int(v) for v in bits.split(',')] for k, bits in _SEL.findall(text)}
        lo = sel.get('op_sel', [0] * len(srcs))
        hi = sel.get('op_sel_hi', [1] * len(srcs))
        swapped = any(s.startswith('v[') and i < len(lo) and lo[i] == 1 for i, s in enumerate(srcs))
        if swapped:
            bad.append((sym, m.group(1) + ' ' + text))
        elif any(s.startswith('v[') and i < len(hi) and hi[i] == 0 for i, s in enumerate(srcs)):
            bcast += 1
    return bad, total, bcast


# Second gate (round 6): kernels whose vector-memory loads must go out in batches.  A load under a per-element branch
# inside an unrolled loop compiles to load / s_waitcnt vmcnt(0) / use, one load in flight per wave -- found in the first
# fused-stem kernel (six serial memory latencies per tile), `gn_stats_kernel`, `sqnorm_partial_kernel` and the RoI gradient
# gather (`c_ok ? load : 0` = seven exec-masked blocks with a wait each per bin row); the fix is an
# unconditional load from a clamped address + a select.  For the kernels below the longest run of global / buffer loads
# with no `s_waitcnt vmcnt` in between must not fall under the listed count again.
MIN_LOADS_IN_FLIGHT = {'stem_pool_kernel': 12, 'gn_stats_kernel': 8, 'sqnorm_partial_kernel': 4,
                       'roi_grad_gather_kernel': 7, 'roi_align_fwd_nhwc_fp_kernel': 7, 'colsum_partial_kernel': 7,
                       'pack_batch_kernel': 4}
_LOAD = re.compile(r'^\s*(global_load_|buffer_load_)(?!.*\blds\b)')
_WAITVM = re.compile(r'^\s*s_waitcnt\b.*vmcnt')


def load_batches(code_object):
    """{kernel symbol: longest run of vector-memory loads issued without waiting for one} of a code object"""
    out = subprocess.run([objdump(), '-d', code_object], check=True, capture_output=True, text=True).stdout
    best, sym, run = {}, None, 0
    for line in out.splitlines():
        m = _SYM.match(line)
        if m:
            sym, run = m.group(1), 0
            best.setdefault(sym, 0)
            continue
        if sym is None:
            continue
        if _LOAD.match(line):
            run += 1
            best[sym] = max(best[sym], run)
        elif _WAITVM.match(line):
            run = 0
    return best


def check_objects(objs, verbose=False, hipcc=None):
    """raise RuntimeError if any object holds a swizzled packed-fp32 instruction; without an llvm-objdump the gate is
    skipped with a warning (the compile itself succeeded: a missing disassembler must not fail the build)"""
    if objdump(hipcc) is None:
        import warnings
        warnings.warn('brcnn build: llvm-objdump not found (looked beside hipcc, in $LLVM_OBJDUMP, PATH, /opt/rocm): '
                      'the packed-fp32 ISA gate was SKIPPED for this build')
        return None
    problems, serial = [], []
    with tempfile.TemporaryDirectory() as wd:
        for obj in objs:
            co = device_code_object(obj, wd)
            bad, total, bcast = swizzled_packed_fp32(co)
            if verbose:
                print(f'{os.path.basename(obj):32s} packed-fp32 instructions {total:6d}  broadcast form {bcast:5d}  cross-swizzled {len(bad)}')
            problems += [(os.path.basename(obj),) + b for b in bad]
            for sym, n in load_batches(co).items():
                for name, need in MIN_LOADS_IN_FLIGHT.items():
                    if name in sym and n < need:
                        serial.append(f'  {os.path.basename(obj)}: {sym}: at most {n} loads in flight, {need} expected')
    if serial:
        raise RuntimeError('vector-memory loads serialised behind a branch (brcnn/check_isa.py MIN_LOADS_IN_FLIGHT; load from a '
                           'clamped address unconditionally and select):\n' + '\n'.join(serial))
    if problems:
        lines = '\n'.join(f'  {o}: {k}: {t}' for o, k, t in problems[:40])
        raise RuntimeError(f'{len(problems)} packed-fp32 instruction(s) with a cross-swizzled VGPR operand (tools/check_isa.py; '
                           f'reshape the source -- scalar fp32 math, or natural-order pairs -- until none is left):\n{lines}')
    return True


def main(argv):
    objdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'obj')
    objs = argv or sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith('.o'))
    if check_objects(objs, verbose=True) is None:
        print('ISA gate skipped: no llvm-objdump')
    else:
        print('ISA gate ok:', len(objs), 'objects')


if __name__ == '__main__':
    main(sys.argv[1:])
