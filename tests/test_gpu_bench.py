"""bench.py on the GPU, as the driver starts it: one short run as a CHILD process under torch.distributed.run with the RCCL path forced
(SRBM_BENCH_FORCE_DIST=1: process group, all-gather of the result records, barriers and the max-over-ranks reduction all execute with a world of
one rank), and the one JSON line it prints carries every field of the measurement contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_line_under_torchrun_with_the_collective_path():
    env = dict(os.environ, SRBM_BENCH_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--repeats', '2', '--gait-steps', '10', '--closed-loop-steps', '4', '--wbc-ticks', '3']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['unit'] == 'it/s' and d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert d['value'] > 1e4 and abs(d['value'] - 256 * 4 / (d['ms_per_step'] * 4 * 1e-3)) < 1e-6 * d['value']
    # the headline: every solve ends by the reference's criterion (lower starting point); the step-rule mode and the library default stand beside it
    assert d['solver_mode'] == 'lower_start' and d['config']['solver']['tol_step'] == 0.0 and d['config']['solver']['start_mu'] > 0
    assert d['config']['solver']['solves_ended_by_step_rule'] == 0
    assert d['step_rule_mode']['solver']['tol_step'] > 0 and d['reference_criterion']['solver']['start_mu'] == 0.0
    assert d['median_region']['value'] >= 0.99 * d['value']
    # the all-gather went through the C-ABI (srbm_allgather_results) with a world of one rank
    assert d['collective']['path'].startswith('C-ABI') and d['collective']['note'] is None and d['rccl_world_size'] == 1
    c = d['config']
    assert 'workload' in c and c['global_batch'] == 256 and c['records_gathered'] == 256 and 'model' not in c
    assert c['timed_solves'] == 256 * 4 * 2 and c['err_bits_all_timed_steps'] == 0
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and 0.05 < rf['frac'] < 1.0
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 10
    assert d['gait']['err_bits_all_steps'] == 0 and d['closed_loop']['plant_finite'] and d['wbc']['finite']
    ss = d['steady_state']        # the headline's protocol continued past the transient after the cold start (bench.py --steady-from)
    assert ss['first_step'] == 125 and ss['all_solved_after'] and len(ss['region_ms']) == 2 and ss['value'] > 1e4


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SRBM_BENCH_FORCE_DIST')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_config_d_full_per_gpu_share_through_the_bench():
    """BASELINE config 4's per-GPU share AT FULL SIZE -- 512 instances, N = 50, push distribution -- through bench.py --workload D as a child
    process (the protocol whose numbers DESIGN.md quotes): every timed solve accounted for through the sticky accumulators, no error bit, and the only solves that do not end Solved belong to the pushed instances whose QPs the ORACLE's solver
    finds primal infeasible as well (tests/test_gpu_parity.py::test_infeasible_qps_of_the_pushed_configuration_are_infeasible_for_the_oracle_too
    exports exactly those QPs and solves them with the restatement)."""
    d = run_bench('--workload', 'D', '--no-cpu-baseline', '--closed-loop-steps', '0')
    c = d['config']
    assert c['batch_per_gpu'] == 512 and c['num_nodes'] == 50 and c['records_gathered'] == 512
    assert c['timed_solves'] == 512 * d['steps'] * d['repeats'] and c['err_bits_all_timed_steps'] == 0 and c['max_iter_in_timed_solves'] == 0
    assert set(c['instances_with_a_solve_not_solved_rank0']) <= {150, 441}, c['instances_with_a_solve_not_solved_rank0']
    assert c['not_solved_in_timed_solves'] <= 2 * d['steps'] * d['repeats']
    assert d['value'] > 4e4, d['value']
    assert d['roofline']['kernel'].startswith('srbm_rti_fused_long')


def test_config_e_full_per_gpu_share_through_the_bench():
    """BASELINE config 5's per-GPU share (128 instances, N = 40: 232 spline variables, the LARGE-capacity build) through bench.py --workload E: every
    timed solve Solved, no error bit"""
    d = run_bench('--workload', 'E', '--no-cpu-baseline', '--closed-loop-steps', '0')
    c = d['config']
    assert c['batch_per_gpu'] == 128 and c['num_nodes'] == 40 and c['records_gathered'] == 128
    assert c['timed_solves'] == 128 * d['steps'] * d['repeats'] and c['all_solved'] and c['err_bits_all_timed_steps'] == 0
    assert '(LARGE build)' in d['roofline']['kernel'] and d['value'] > 5e3
