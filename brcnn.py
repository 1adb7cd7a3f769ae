"""Import alias: `import brcnn` loads the package directory `boosting-r-cnn_amd/` (whose name
is not a valid Python identifier) and installs it in sys.modules as `brcnn`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'boosting-r-cnn_amd')
_spec = importlib.util.spec_from_file_location(
    'brcnn', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['brcnn'] = _mod
_spec.loader.exec_module(_mod)
