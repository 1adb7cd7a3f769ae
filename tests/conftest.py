import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _host_cpus():
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return n


# torch's intra-op pool would take one thread per core of the host (256 on the GPU boxes) inside a container with a
# 16-CPU quota: every CPU-side parallel region (weight initialisation, the CPU chains the parity tests compare against)
# gets the process throttled for the rest of its 100 ms scheduling period.  Read by OpenMP when torch is first imported.
os.environ.setdefault('OMP_NUM_THREADS', str(_host_cpus()))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# Collection order of the -m gpu suite: the parity tests proper (HIP path against the C oracle and against the golden
# fixtures of the imported reference) run FIRST, file by file in the order below; kernel self-consistency and stress
# files next; the multi-process integration files (launchers, torch.distributed.run children) LAST.  Under `pytest -x`
# one launcher flake must not keep a single oracle / golden comparison from running (round 4: an integration test that
# sorted alphabetically in front of every parity file stopped the driver's run before any of them).
_ORDER = ['test_ops_gpu', 'test_golden_gpu', 'test_fullsize_gpu', 'test_train_gpu', 'test_f16_gpu', 'test_bf16_gpu',
          'test_fuzz_gpu', 'test_stress_gpu', 'test_drivers_gpu', 'test_ddp_gpu']


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 2     # unknown files: before the integration files


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)           # stable: the order inside a file stays the file's own
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
