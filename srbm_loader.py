"""Imports bilevel-gait-gen_amd/host.py (the directory name is not a Python identifier)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location('srbm_host', os.path.join(ROOT, 'bilevel-gait-gen_amd', 'host.py'))
host = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(host)
