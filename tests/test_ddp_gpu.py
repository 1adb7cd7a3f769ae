"""-m gpu: the multi-GPU path on the hardware at hand (one MI355X): world-size-1 `nccl` (= RCCL) runs of
(a) one train step under DistributedDataParallel against the unwrapped step, fp32 and bf16,
(b) tools/train.py --launcher pytorch (train_detector(distributed=True): DistributedGroupSampler, DDP,
    DistSamplerSeedHook, multi_gpu_test) for one epoch, and
(c) bench.py launched through torch.distributed.run as the driver launches it for N > 1.
World size 2 runs on CPU (gloo) in tests/test_distributed_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.test_drivers_cpu import _tiny_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _stalled(r):
    """the run ended because the watchdog fired (every thread's stack on stderr), not because an assertion failed"""
    return r is None or (r.returncode != 0 and 'Timeout (' in r.stderr and 'AssertionError' not in r.stderr and
                         'Error' not in r.stderr.replace('ChildFailedError', ''))


def _launch(script_args, timeout=900, nproc=1, extra_env=None, stall_s=None):
    """`stall_s`: for the legs that keep gloo's staged device<->host copies in flight UNDER the backward pass of two
    processes sharing one GPU (the overlapped reducer on this harness only: RCCL does not stage through the host).  That
    combination sometimes crawls -- 15-60 s per step instead of 1.3 s, both ranks at the same collective, measured in
    profiles/r05_notes.md.  A crawl is random, a deadlock (an unpaired or mis-ordered collective: how the overlapped
    reducer would fail) is not: a run whose watchdog fires after `stall_s` seconds is REPEATED once, and a second stall
    FAILS the test (VERDICT r05 item 4c / ADVICE r05: a hang must not read as a skip).  BRCNN_ALLOW_HARNESS_STALL=1
    turns the double stall back into a skip for a box where the harness is known to crawl."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr',
           '127.0.0.1', '--master-port', str(_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, **(extra_env or {}))
    if stall_s is None:
        return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    env['DDP_WATCHDOG_S'] = env['BRCNN_WATCHDOG_S'] = str(int(stall_s))
    tails = []
    for attempt in range(2):
        cmd[cmd.index('--master-port') + 1] = str(_port())
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=stall_s + 60)
        except subprocess.TimeoutExpired as e:
            r = None
            tails.append('(killed by the launcher timeout) ' + str(e.stderr or '')[-1500:])
        if not _stalled(r):
            return r
        if r is not None:
            tails.append(r.stderr[-1500:])
    msg = (f'the two-rank run stalled for {stall_s} s TWICE in a row (watchdog fired, no assertion failed): a crawl of the '
           'one-GPU gloo harness (profiles/r05_notes.md) does not repeat like that -- treat as a deadlock of the overlapped '
           'gradient exchange.  stderr tails of the two attempts:\n' + '\n-----\n'.join(tails))
    if os.environ.get('BRCNN_ALLOW_HARNESS_STALL') == '1':
        pytest.skip(msg)
    pytest.fail(msg)


# two ranks on the one GPU of the box: RCCL refuses that, so the collectives go through gloo (device tensors staged
# through the host) -- the world-size-2 control flow of the product code on real kernels
TWO_ON_ONE = dict(BRCNN_DIST_ONE_DEVICE='1', BRCNN_DIST_BACKEND='gloo')


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_ddp_train_step_equals_plain_step(dtype):
    r = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py'), dtype])
    assert r.returncode == 0 and 'DDP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_two_rank_train_step_reducer_and_ddp_agree(dtype):
    """world size 2 (both ranks on cuda:0, gloo): DistributedDataParallel and GradReducer (the RPN branch back-propagated
    inside the forward pass or not, weights as loaded or channels-last) deliver the mean of the two ranks' unwrapped
    gradients, and agree with each other"""
    r = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py'), dtype, 'ddp,own'], nproc=2, extra_env=TWO_ON_ONE)
    assert r.returncode == 0 and 'DDP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_two_rank_overlapped_reducer_delivers_the_mean(dtype):
    """the overlapped form at world size 2 (both ranks on cuda:0, gloo): arena slices all-reduced IN PLACE while the
    backward pass runs, only where the slice is `.grad` itself -- {early RPN backward off, on} x {weights as loaded,
    channels-last}, every gradient equal to the mean of the two ranks' unwrapped gradients"""
    r = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py'), dtype, 'overlap'], nproc=2, extra_env=TWO_ON_ONE, stall_s=240)
    assert r.returncode == 0 and 'DDP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_rank_overlap_with_stalled_main_stream():
    """the overlapped reducer with the main stream held back 6 ms behind every weight-gradient launch: autograd's
    main-stream copies of a weight gradient then run AFTER an in-place all-reduce of the same arena slice would have
    completed.  Round 4 sliced the arena by offset alone and reduced such a copy a second time (intermittently: the
    driver's red run); slices now hold only ranges that ARE `.grad` (distributed.GradReducer.writers_launched)."""
    r = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py'), 'f32', 'overlap'], nproc=2,
                extra_env=dict(TWO_ON_ONE, DDP_WORKER_STALL_MS='6'), stall_s=300)
    assert r.returncode == 0 and 'DDP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_train_tool_distributed_launcher(tmp_path):
    cfg = _tiny_cfg(tmp_path, max_epochs=1)
    cfg_path = str(tmp_path / 'tiny_cfg.py')
    cfg.dump(cfg_path)
    work = str(tmp_path / 'work_ddp')
    r = _launch([os.path.join(ROOT, 'tools', 'train.py'), cfg_path, '--work-dir', work, '--seed', '0', '--launcher',
                 'pytorch', '--allow-random-init'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert os.path.exists(os.path.join(work, 'epoch_1.pth'))


def test_train_tool_two_ranks_keep_identical_replicas(tmp_path):
    """tools/train.py --launcher pytorch on two ranks (both on cuda:0, gloo): DistributedGroupSampler shards, the
    GradReducer averages, and after EVERY optimizer step the two replicas hold bit-identical parameters and buffers
    (runner.check_replicas: a checksum compared across the ranks)"""
    cfg = _tiny_cfg(tmp_path, max_epochs=1)
    cfg.check_replicas = True
    cfg_path = str(tmp_path / 'tiny_cfg.py')
    cfg.dump(cfg_path)
    work = str(tmp_path / 'work_two')
    r = _launch([os.path.join(ROOT, 'tools', 'train.py'), cfg_path, '--work-dir', work, '--seed', '0', '--launcher',
                 'pytorch', '--allow-random-init'], nproc=2, extra_env=TWO_ON_ONE, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert os.path.exists(os.path.join(work, 'epoch_1.pth'))
    out = r.stdout + r.stderr
    assert 'replica check: parameters and buffers bit-identical on every rank after each of' in out, out[-3000:]


def test_bench_under_the_launcher():
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2',
                 '--no-cpu-baseline'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['train']['value'] > 0 and 'roofline' in line
    # a world size that contradicts --gpus is refused instead of silently measuring one GPU
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stdout + r.stderr)


def test_bench_two_ranks_prints_one_line_with_reduce_time():
    """`bench.py --gpus 2` as the driver launches it, both ranks on the one GPU (gloo): every rank runs the same
    collectives -- the roofline pass included -- and rank 0 prints ONE line carrying train.reduce_ms / grad_bytes"""
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
                 '--no-cpu-baseline'], nproc=2, extra_env=TWO_ON_ONE, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 4 and line['value'] > 0
    tr = line['train']
    assert tr['n_gpus'] == 2 and tr['reduce_ms'] > 0 and tr['grad_bytes'] > 150e6 and tr['roofline']['frac'] > 0
    assert tr['grad_allreduce']['world'] == 2 and tr['grad_allreduce']['bytes_last_step'] >= tr['grad_bytes'] * 0.9


def test_bench_two_ranks_overlapped_and_bf16_wire_format():
    """the two other forms of the gradient exchange through `bench.py --gpus 2` (both ranks on the one GPU, gloo):
    overlapped in-place slices (BRCNN_REDUCER_OVERLAP=1) and the bf16 wire format (BRCNN_REDUCER_COMPRESS=bf16); every
    N line carries reduce_ms / grad_bytes / what was exchanged"""
    # (the default 64 MiB slices: with many small slices in flight this one-GPU gloo harness crawls, profiles/r05_notes.md)
    for env, check, stall in ((dict(BRCNN_REDUCER_COMPRESS='bf16'), lambda d: 'bf16' in d['arena'], None),
                              (dict(BRCNN_REDUCER_OVERLAP='1'), lambda d: d['overlap'] is True, 240)):
        r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
                     '--mode', 'train', '--no-cpu-baseline'], nproc=2, extra_env=dict(TWO_ON_ONE, **env), timeout=1500,
                    stall_s=stall)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1
        line = json.loads(lines[0])
        assert line['n_gpus'] == 2 and line['value'] > 0 and line['reduce_ms'] > 0 and line['grad_bytes'] > 150e6
        assert check(line['grad_allreduce']), line['grad_allreduce']
        assert line['step_ms_median'] > 0 and line['step_ms_max'] >= line['step_ms_median']
