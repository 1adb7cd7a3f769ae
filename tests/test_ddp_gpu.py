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


def _launch(script_args, timeout=900):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_ddp_train_step_equals_plain_step(dtype):
    r = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py'), dtype])
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


def test_bench_under_the_launcher():
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2',
                 '--no-cpu-baseline'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['train']['value'] > 0 and 'roofline' in line
    # a world size that contradicts --gpus is refused instead of silently measuring one GPU
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stdout + r.stderr)
