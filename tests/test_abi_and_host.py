"""CPU-side checks of the product: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/srbm_rti.h declares (no compute calls without a GPU); host-side helpers; multi-rank sharding logic (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from srbm_loader import host, ROOT


@pytest.fixture(scope='module')
def libpath():
    return host.build()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'srbm_rti.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(srbm_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(libpath):
    names = declared_symbols()
    assert len(names) >= 60
    for path in (libpath, host.LIB_PATH_LARGE):        # the standard and the LARGE-capacity build carry the same C-ABI
        L = ctypes.CDLL(path)
        for n in names:
            assert hasattr(L, n), (os.path.basename(path), n)
    cap = (ctypes.c_int * 4)()
    ctypes.CDLL(libpath).srbm_get_capacity(cap)
    assert list(cap) == [50, 160, 120, 32]
    ctypes.CDLL(host.LIB_PATH_LARGE).srbm_get_capacity(cap)
    assert list(cap) == [100, 240, 200, 32]


def test_code_object_is_gfx950(libpath):
    """single-target build: every offload target named in the fat binary is CDNA4 (gfx950)"""
    blob = open(libpath, 'rb').read()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_no_gpu_fails_loudly(libpath):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    cfg = host.load_config()
    with pytest.raises(RuntimeError):
        host.BatchMPC(cfg, 2)


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, 'bilevel-gait-gen_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hiph', '.h', '.cpp', 'Makefile')):
                txt = open(os.path.join(dirpath, f), errors='ignore').read()
                assert 'oracle' not in txt.lower() or f == 'host.py' and 'oracle' not in txt, os.path.join(dirpath, f)


def test_manifold_tangent_helpers():
    q = np.array([0.0505, -0.1643, -0.0572, 0.9835]); q /= np.linalg.norm(q)
    s = np.concatenate([[0, 0, 0.3], [1, 2, 3], q, [0.1, 0.2, 0.3]])
    t = host.manifold_to_tangent(s)
    th = np.linalg.norm(t[6:9])
    assert abs(np.cos(th / 2) - q[3]) < 1e-12 and np.allclose(np.sin(th / 2) * t[6:9] / th, q[:3], atol=1e-12)


def test_config_files_carry_the_reference_values():
    c = host.load_config('a1_configuration')
    assert (c['num_nodes'], c['integrator_dt'], c['friction_coef'], c['force_bound']) == (20, 0.05, 0.5, 150)
    assert abs(c['mass'] - 13.741) < 1e-9        # sum of <mass> in models/a1_description/urdf/a1.urdf
    assert np.allclose(np.abs(np.array(c['hip_xy'])), [[0.1805, 0.047]] * 4)
    d = host.load_config('a1_config_distr_rejection')
    assert (d['num_nodes'], d['integrator_dt'], d['force_cost'], d['gait_opt_freq']) == (50, 0.02, 0.001, 5)


def _run_two_ranks(code):
    port = str(29500 + os.getpid() % 400)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)


def test_sharding_two_ranks_gloo():
    """bench.py shards instances over ranks with no data-path collective; results are collected with one all_gather.
    World size 2 on CPU (gloo): the shard arithmetic and the gather of result records."""
    _run_two_ranks(r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lo, hi = bench.shard_range(10, rank, world)
assert (lo, hi) == ((0, 5) if rank == 0 else (5, 10)), (lo, hi)
rec = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 4)
out = bench.gather_records(rec, world)
assert out.shape == (10, 4) and torch.equal(out[:, 0], torch.arange(10, dtype=torch.float64))
t = bench.max_over_ranks(float(rank + 1))
assert t == 2.0
dist.barrier(); dist.destroy_process_group()
''' % ROOT)


def test_bench_dry_run_eight_ranks_config_d():
    """BASELINE config 4 at its full size: `bench.py --gpus 8 --workload D --dry-run` spawns 8 ranks (gloo), every rank owns exactly 512 of the
    4096 instances (contiguous, exact bounds) and rank 0 holds all 4096 records after the all-gather"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dry-run', '--workload', 'D'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 8 and d['collective_world_size'] == 8 and d['scaling'] == 'weak'
    assert c['batch_per_gpu'] == 512 and c['global_batch'] == 4096 and c['records_gathered'] == 4096 and c['gather_ok']
    assert c['shard_bounds'] == [[512 * r_, 512 * (r_ + 1)] for r_ in range(8)]


def test_build_is_serialised_by_a_file_lock():
    """host.build() of two processes at once: one `make` at a time on the tree (the ranks of a multi-GPU run on a fresh box).  With the libraries
    present build() returns without running make; force=True takes the lock -- two forced builds in parallel both succeed."""
    code = 'import sys; sys.path.insert(0, %r); from srbm_loader import host; host.build(force=True); print("built")' % ROOT
    procs = [subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all('built' in o for o in outs), outs
    assert os.path.exists(host.LIB_PATH) and os.path.exists(host.LIB_PATH_LARGE)


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` outside torchrun must produce TWO ranks (the launcher spawns fresh processes before anything
    touches a GPU); --dry-run walks the launcher, the sharding and the all-gather on CPU (gloo) without a HIP call."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    import json
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 512 and d['config']['records_gathered'] == 512 and d['config']['gather_ok']
    # the same for the per-GPU share of BASELINE config 4 (512 instances x 2 ranks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--workload', 'D'], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['config']['batch_per_gpu'] == 512 and d['config']['global_batch'] == 1024 and d['config']['records_gathered'] == 1024 and d['config']['gather_ok']
    # a world size that contradicts --gpus is refused, never reported as n_gpus = 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE='1', RANK='0'), timeout=300)
    assert r.returncode != 0 and 'refusing' in r.stderr


def test_trajectory_record_layout_and_host_evaluation(libpath):
    """srbm_trajectory (mpc::Trajectory as a flat record): the ctypes mirror has the library's layout, and
    Trajectory::GetForce / GetEndEffectorLocation / GetContacts evaluate on the HOST (no GPU) with the reference's known
    answers of test/splines_tests.cpp:58-107 (5 contact times 0, 0.2, ..., 0.8, the foot starting in swing)."""
    L = ctypes.CDLL(libpath)
    assert L.srbm_sizeof_trajectory() == ctypes.sizeof(host.Trajectory)
    t = host.Trajectory()
    t.num_states = 21; t.node_dt = 0.05; t.swing_height = 6.0; t.foot_offset = 0.0
    # knot pattern of a swing-first foot (end_effector_splines.cpp:45-100): LO mid TD F F LO mid TD F F LO
    times = [0.0, 0.1, 0.2, 0.2 + 0.2 / 3, 0.2 + 0.4 / 3, 0.4, 0.5, 0.6, 0.6 + 0.2 / 3, 0.6 + 0.4 / 3, 0.8]
    kinds = [0, 3, 1, 2, 2, 0, 3, 1, 2, 2, 0]
    for ee in range(4):
        t.nk[ee] = len(times)
        for k, (tt, kd) in enumerate(zip(times, kinds)):
            t.knot_time[ee][k] = tt; t.knot_kind[ee][k] = kd
    # position x as the reference test sets it (value = index of the mutable node, :60-62): 0 on the first lift-off knot, 2 through
    # the first stance (TD knot 2 and its LO knot 5), 7 through the second (knots 7 and 10)
    for k, v in zip([0, 2, 5, 7, 10], [0.0, 2.0, 2.0, 7.0, 7.0]):
        t.pos_xy[0][0][k] = v
    assert t.get_end_effector_location(0, 0.0)[0] == 0.0
    assert abs(t.get_end_effector_location(0, 0.103448)[0] - 1.0517) < 1e-3       # test/splines_tests.cpp:67
    assert abs(t.get_end_effector_location(0, 0.503448)[0] - 4.62926) < 1e-3      # :68
    assert abs(t.get_end_effector_location(0, 0.5)[2] - 6.0) < 1e-12         # z at the mid-swing knot = swing height (:79-81)
    assert t.get_contacts(0.3) == [True] * 4 and t.get_contacts(0.1) == [False] * 4
    assert t.get_contact_times()[0] == [0.0, 0.2, 0.4, 0.6, 0.8]
    with pytest.raises(RuntimeError):
        t.get_force(0, -5.0)                   # before the first knot: the reference throws


def test_reference_yaml_configuration_loads(tmp_path):
    """host.load_yaml_config reads the reference's own YAML keys (test/simulation_mpc.cpp:55-89); checked on a YAML written
    from the committed JSON configuration (the reference's files do not travel)"""
    import json, yaml
    from srbm_loader import host
    cfg = host.load_config('a1_configuration')
    y = {k: cfg[k] for k in ['num_nodes', 'integrator_dt', 'friction_coef', 'force_bound', 'swing_height', 'foot_offset', 'ee_box_size',
                             'force_cost', 'Q_srbd_diag', 'srb_init', 'srb_target', 'gait_opt_freq']}
    p = tmp_path / 'a1.yaml'
    p.write_text(yaml.safe_dump(y))
    c2 = host.load_yaml_config(str(p), dict(mass=cfg['mass'], Ir=cfg['Ir'], hip_xy=cfg['hip_xy']))
    for k in y:
        assert c2[k] == cfg[k], k
    assert np.allclose(c2['Ir'], cfg['Ir']) and np.allclose(c2['hip_xy'], cfg['hip_xy'])
    (tmp_path / 'bad.yaml').write_text(yaml.safe_dump({'num_nodes': 20}))
    with pytest.raises(KeyError):
        host.load_yaml_config(str(tmp_path / 'bad.yaml'), dict(mass=1, Ir=np.eye(3), hip_xy=np.zeros((4, 2))))
