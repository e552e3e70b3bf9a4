"""The C++ host side above the C-ABI (include/srbm_rti.hpp: srbm::MPCSingleRigidBody, srbm::GaitOptimizer -- the reference's
class and method names).  CPU: the header and the smoke program compile with g++ and link against libsrbm_rti.so.
GPU: the program's read-backs equal the ctypes path (same library, same call sequence: bit-identical)."""
import json
import os
import subprocess

import numpy as np
import pytest

from srbm_loader import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, 'tests', 'cpp')


def write_cfg_inc(cfg, path):
    tgt = np.array(cfg['srb_target'], float)
    tt = host.manifold_to_tangent(tgt)
    arr = lambda name, v: 'static const double %s[%d] = {%s};\n' % (name, len(v), ', '.join(repr(float(x)) for x in v))
    with open(path, 'w') as f:
        f.write('static const int kNumNodes = %d;\n' % cfg['num_nodes'])
        for name, key in [('kDt', 'integrator_dt'), ('kMu', 'friction_coef'), ('kForceBound', 'force_bound'), ('kSwing', 'swing_height'),
                          ('kFootOffset', 'foot_offset'), ('kForceCost', 'force_cost'), ('kMass', 'mass')]:
            f.write('static const double %s = %r;\n' % (name, float(cfg[key])))
        f.write(arr('kBox', cfg['ee_box_size'])); f.write(arr('kIr', np.array(cfg['Ir']).reshape(-1)))
        f.write(arr('kHip', np.array(cfg['hip_xy']).reshape(-1))); f.write(arr('kQdiag', cfg['Q_srbd_diag']))
        f.write(arr('kInit', cfg['srb_init'])); f.write(arr('kTarget13', tgt)); f.write(arr('kTargetTangent', tt))


def build_program(tmpdir):
    cfg = host.load_config('a1_configuration')
    host.build()
    write_cfg_inc(cfg, os.path.join(tmpdir, 'cfg.inc'))
    exe = os.path.join(tmpdir, 'facade_smoke')
    libdir = os.path.dirname(host.LIB_PATH)
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-I', tmpdir,
                           os.path.join(CPP, 'facade_smoke.cpp'), '-o', exe, '-L', libdir, '-lsrbm_rti', '-Wl,-rpath,' + libdir])
    return cfg, exe


def test_cpp_host_side_compiles_and_links(tmp_path):
    cfg, exe = build_program(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_host_side_equals_ctypes_path(tmp_path):
    cfg, exe = build_program(str(tmp_path))
    out = subprocess.check_output([exe], text=True)
    lines = out.strip().splitlines()
    head = lines[0].split()
    vals = {}
    for ln in lines[1:]:
        k, i, v = ln.split()
        vals.setdefault(k, []).append(float(v))
    # the same sequence through ctypes
    s0 = np.array(cfg['srb_init'], float)
    ee0 = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
    g = host.BatchMPC(cfg, 2)
    g.set_state_trajectory_warm_start(s0)
    g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    g.create_initial_run(s0, ee0)
    g.rti_advance(0, 4); g.synchronize()
    gait = host.BatchGaitOptimizer(g)
    gait.compute_gradient()
    grad, valid = gait.gradient()
    t = 4 * cfg['integrator_dt']
    gait.optimize_contact_times(t)
    step = gait.step()
    st1 = g.trajectory_states()[:, 1, :]
    imin, costs = gait.line_search(st1, t, ee0)
    st, err = g.status()
    sz = g.sizes()
    x = g.qp_solution()
    assert [int(v) for v in head[1:3]] == [int(st[0]), int(st[1])]
    assert int(head[7]) == sz[0, 0] and int(head[9]) == sz[0, 1] and int(head[11]) == valid[0] and int(head[13]) == imin[0]
    assert np.array_equal(np.array(vals['grad']), grad[0, :20])
    assert np.array_equal(np.array(vals['step']), step[0, :20])
    assert np.array_equal(np.array(vals['x']), x[0, :40])
