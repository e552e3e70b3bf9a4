"""Diagnostic: duration of the stand-alone condensing kernel (host timer around 50 host-driven RTI steps is too coarse; this
uses the unfused path, which launches one kernel per phase, under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
import bench
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
g.create_initial_run(states, ees)
g.rti_advance_unfused(0, 20); g.synchronize()
