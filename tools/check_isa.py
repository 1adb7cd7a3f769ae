"""command-line front of the build's ISA gate (the gate itself lives in the package: boosting-r-cnn_amd/check_isa.py,
so that a build triggered by `lib.load()` does not depend on this directory): `python tools/check_isa.py [objects...]`"""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'boosting-r-cnn_amd', 'check_isa.py')
_spec = importlib.util.spec_from_file_location('brcnn_check_isa', _p)
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith('__')})

if __name__ == '__main__':
    _m.main(sys.argv[1:])
