"""Multi-step launches of a batch LARGER than the chip run on the step queues (srbm_fused.hiph: srbm_rti_queued): a resident grid takes
(instance, step) items instead of one workgroup walking one instance through all steps.  The arithmetic of a step is the same, phase by
phase, so the results must be BITWISE those of the one-workgroup-per-instance launch (SRBM_NO_STEP_QUEUE=1), whatever the order in which
the items were taken -- checked here on Config D's instances (N = 50, pushes under a plant in the second test) in child processes, one per
setting of the switch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from srbm_loader import host, workloads
cfg = host.load_config('a1_config_distr_rejection')
B = int(sys.argv[1]); steps = int(sys.argv[2]); closed = int(sys.argv[3]); rewind = int(sys.argv[4])
st, ee = zip(*[workloads.config_d_instance(cfg, b %% 512) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_step_rule(float(os.environ.get('QUEUE_TEST_TOL_STEP', '0')), 0.1)
g.create_initial_run(st, ee)
if closed:
    g.plant_set_state(st)
    imp = np.zeros((B, 6)); imp[::7, 0] = 0.4; imp[::11, 1] = -0.3
    g.plant_set_push(time=np.full(B, 2.5 * cfg['integrator_dt']), impulse=imp)
    g.closed_loop_advance(0, steps)
else:
    g.rti_advance(0, steps)
    if rewind:
        # the protocol of scripts/dev_ab.py: on after the first launch, then BACK to an earlier index: the instances whose plan has moved on raise time
        # errors and run into the iteration limit at every step -- a few instances 20 x slower than the rest, the workgroups that hold the
        # last tickets of a queue wait for ALL their remaining steps (the case the restartable timeout of the queue is written for)
        g.rti_advance(steps, rewind)
        g.rti_advance(steps, 40)
g.synchronize()
acc = g.status_accumulated()
out = dict(x=g.qp_solution().tolist(), states=g.trajectory_states().tolist(), status=g.status()[0].tolist(), err=int(np.bitwise_or.reduce(acc[:, 0])),
           solves=acc[:, 1].tolist(), iters=g.work_counters()[0])
print('RESULT ' + json.dumps(out))
'''


def run(batch, steps, closed, no_queue, rewind=0, tol_step=0.0):
    env = dict(os.environ)
    env.pop('SRBM_NO_STEP_QUEUE', None)
    if no_queue:
        env['SRBM_NO_STEP_QUEUE'] = '1'
    env['QUEUE_TEST_TOL_STEP'] = repr(tol_step)
    p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT), str(batch), str(steps), str(closed), str(rewind)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[7:])


def same(a, b, err_allowed=0):
    assert (a['err'] & ~err_allowed) == 0 and a['err'] == b['err'], (a['err'], b['err'])
    assert a['status'] == b['status']
    assert a['solves'] == b['solves'] and a['iters'] == b['iters']
    xa, xb = np.array(a['x']), np.array(b['x'])
    sa, sb = np.array(a['states']), np.array(b['states'])
    assert np.array_equal(np.nan_to_num(xa), np.nan_to_num(xb)), float(np.nanmax(np.abs(xa - xb)))
    assert np.array_equal(sa, sb), float(np.abs(sa - sb).max())


def test_queued_launch_is_bitwise_the_per_instance_launch():
    # 600 instances (not a multiple of 8 x anything: ragged queues), 6 steps: 3 600 items over the resident grid
    q = run(600, 6, 0, False)
    r = run(600, 6, 0, True)
    assert len(set(q['solves'])) == 1 and q['solves'][0] >= 7        # every instance went through the cold start and all six steps
    same(q, r)


def test_queued_closed_loop_with_pushes_is_bitwise_the_per_instance_launch():
    q = run(520, 5, 1, False)
    r = run(520, 5, 1, True)
    same(q, r)


def test_queued_launch_with_a_few_very_slow_instances_is_bitwise_the_per_instance_launch():
    # time errors (bits 1, 4) are the protocol's doing and the same in both launches; the queue's own bit (512) must stay clear
    q = run(512, 5, 0, False, rewind=100, tol_step=1e-5)
    r = run(512, 5, 0, True, rewind=100, tol_step=1e-5)
    same(q, r, err_allowed=1 | 2 | 4)
