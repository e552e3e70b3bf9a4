"""Developer script: time the device-resident RTI protocol (no oracle)."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('srbm_host', os.path.join(ROOT, 'bilevel-gait-gen_amd', 'host.py'))
host = importlib.util.module_from_spec(spec); spec.loader.exec_module(host)
cfg = host.load_config(sys.argv[1] if len(sys.argv) > 1 else 'a1_configuration')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
s0 = np.array(cfg['srb_init'], float)
ee = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
gb = host.BatchMPC(cfg, B); gb.set_state_trajectory_warm_start(s0); gb.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
gb.create_initial_run(s0, ee)
gb.rti_advance(0, 3); gb.synchronize()
t0 = time.time()
gb.rti_advance(3, K); gb.synchronize()
el = time.time() - t0
st, err = gb.status()
print('batch %d: %d RTI steps in %.4f s -> %.1f it/s (%.3f ms/step) status %s err %s iters %.1f' % (B, K, el, B * K / el, 1e3 * el / K, np.unique(st), np.unique(err), gb.stats()[:, 4].mean()))
