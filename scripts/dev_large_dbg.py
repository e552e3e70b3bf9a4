import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from srbm_loader import host
from bench import config_b_instance
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = host.load_config('a1_configuration', num_nodes=N)
B = 8
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
for fused in (False, True):
    g = host.BatchMPC(cfg, B, large=True)
    g.set_state_trajectory_warm_start(states)
    g.create_initial_run(states, ees)
    for i in range(0, 12, 2):
        (g.rti_advance if fused else g.rti_advance_unfused)(i, 2); g.synchronize()
        st, err = g.status(); sz = g.sizes()
        print('fused' if fused else 'unfused', 'steps', i + 2, 'status', st.tolist(), 'err', err.tolist(), 'n', sz[0, 0], 'samples', sz[0, 7], 'nk', g.knots(0)['nk'])
