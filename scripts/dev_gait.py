import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from test_gpu_gait import run_pair, host
cfg, g, o, state, ee, t = run_pair('a1_configuration', 3)
grad = o.gait_gradient(); step, new_times = o.gait_optimize(t); nv = len(grad)
print('step', step[:nv])
gait = host.BatchGaitOptimizer(g); gait.set_contact_times_from_trajectory(); gait.set_step(step[:nv])
k_o, costs_o = o.gait_line_search(state, t, ee)
imin, costs = gait.line_search(state, t, ee)
print('imin', imin, k_o); print(costs[0]); print(costs_o)
kg = g.knots(0)
for e in range(4):
    ko = o.knots(e)
    K = ko['K']
    print(e, kg['nk'][e], K, (kg['times'][e, :K] - ko['times']))
    print('   ', kg['times'][e, :K]); print('   ', ko['times'])
