"""Do the device and the numpy restatement agree on WHICH target solves fail?  Config B batch after 7 RTI steps (a state in which a fifth of
the inverse-kinematics solves hit the iteration limit), targets at init_time + 1 ms: statuses and, for a few failing instances, the iterates."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from srbm_loader import host
import bench
import ik_numpy as ik
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
g.create_initial_run(states, ees)
g.rti_advance(0, int(sys.argv[1]) if len(sys.argv) > 1 else 7); g.synchronize()
trajs = g.get_trajectory()
legs = np.array(cfg['leg_origins']); q0 = np.array(cfg['init_config'], float)
Ir_inv = np.linalg.inv(np.array(cfg['Ir']))
t = trajs[0].init_time + 1e-3
q, v, f, st = g.get_targets_from_traj(t, np.tile(q0, (B, 1)))
print('device statuses', dict(zip(*np.unique(st, return_counts=True))))
bad = np.where(st != 0)[0]
check = list(bad[:6]) + list(np.where(st == 0)[0][:3])
agree = 0
for b in check:
    tr = trajs[b]
    qo, vo, fo, ok = ik.targets_from_traj(legs, tr.get_states(), tr.init_time, tr.node_dt, cfg['mass'], Ir_inv,
                                          lambda e, tt: tr.get_end_effector_location(e, tt), lambda e, tt: tr.get_force(e, tt), t, q0)
    same = (ok and st[b] == 0) or ((not ok) and st[b] == 1)
    agree += same
    print('instance %3d device status %d oracle ok %s  |dq| %.2e' % (b, st[b], ok, np.abs(q[b] - qo).max()))
    if st[b] != 0:
        _, pos, con = g.eval_trajectory(t)
        s = tr.get_states()
        print('    base z %.3f  foot targets z %s  xy offsets from base %s' % (s[0][2], np.round(pos[b][:, 2], 3), np.round(pos[b][:, :2] - s[0][:2], 3).tolist()))
print('agreement %d / %d' % (agree, len(check)))
