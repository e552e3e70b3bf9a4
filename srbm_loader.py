"""Imports the modules of bilevel-gait-gen_amd/ (the directory name is not a Python identifier): `host` (ctypes binding of the C-ABI) and
`workloads` (seeded instance generators of the BASELINE configurations, sharding)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'bilevel-gait-gen_amd', file))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


host = _load('srbm_host', 'host.py')
workloads = _load('srbm_workloads', 'workloads.py')
sys.modules[__name__ + '.workloads'] = workloads          # `from srbm_loader.workloads import ...`
sys.modules[__name__ + '.host'] = host
