"""bench.py on the GPU, as the driver starts it: one short run as a CHILD process under torch.distributed.run with the RCCL path forced
(SRBM_BENCH_FORCE_DIST=1: process group, all-gather of the result records, barriers and the max-over-ranks reduction all execute with a world of
one rank), and the one JSON line it prints carries every field of the measurement contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_under_torchrun_with_the_collective_path():
    env = dict(os.environ, SRBM_BENCH_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29577',
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
