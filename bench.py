#!/usr/bin/env python3
"""Headline benchmark: MPC RTI iterations / second, batched A1 SRBM, N=20 (BASELINE.json `metric`, config[1]).

Workload (Config B): 256 independent MPC instances PER GPU, horizon N=20, dt=0.05,
apps/a1_configuration.yaml values, synthetic initial states (std::mt19937_64-style seeds 20240112+b, here numpy MT19937).
Protocol: 10 cold-start solves per instance (MPC::CreateInitialRun, untimed set-up), W warm-up RTI steps, then K timed
RTI steps with t_i = i*dt and state := node 1 of the previous trajectory (test/gait_opt_playground.cpp:113-126), run
device-resident (inputs are in HBM when the timed region starts).  One "step" = one RTI iteration of every instance
of the batch (shift -> assemble -> condense -> QP solve -> line search -> trajectory update).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); instances are sharded over ranks (weak
scaling: 256 per GPU) with no data-path collective; one all-gather of the per-instance result records closes the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (public spec; the CDNA guide lists no fp64 row)
BATCH_PER_GPU = 256
RESULT_LD = 8 + 21 * 12 + 160


# ---------- helpers shared with the CPU (gloo) test of the sharding logic ----------
def shard_range(total, rank, world):
    """contiguous block [lo, hi) of `total` instances owned by `rank`"""
    per = total // world
    rem = total % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def gather_records(rec, world):
    """all-gather of fixed-size per-instance result records (rows) over the ranks"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return rec
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def max_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def config_b_instance(cfg, b):
    """instance b of Config B: perturbed initial state + foot positions"""
    rng = np.random.Generator(np.random.MT19937(20240112 + b))
    u = lambda lo, hi: lo + (hi - lo) * rng.random()
    m = cfg['mass']
    p = np.array([u(-0.02, 0.02), u(-0.02, 0.02), 0.30 + u(-0.01, 0.01)])
    v = np.array([u(-0.5, 0.5), u(-0.5, 0.5), u(-0.1, 0.1)])
    rpy = np.array([u(-0.05, 0.05), u(-0.05, 0.05), u(-0.05, 0.05)])
    L = np.array([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)])
    th = np.linalg.norm(rpy)
    quat = np.concatenate([np.sin(th / 2) / th * rpy, [np.cos(th / 2)]])
    state = np.concatenate([p, m * v, quat, L])
    hips = np.array([[0.2055, 0.147], [0.2055, -0.147], [-0.1555, 0.147], [-0.1555, -0.147]])
    ee = np.zeros((4, 3))
    for e in range(4):
        ee[e, 0] = p[0] + hips[e, 0] + u(-0.02, 0.02)
        ee[e, 1] = p[1] + hips[e, 1] + u(-0.02, 0.02)
    return state, ee


def config_d_instance(cfg, b):
    """instance b of Config D: apps/a1_config_distr_rejection.yaml values (N=50, dt=0.02); the file's
    single push becomes a distribution -- lin-mom xy ~ N(0, 2.5^2) truncated at 3 sigma, yaw ang-mom ~ N(0, 0.2^2), seed 777 + b"""
    rng = np.random.Generator(np.random.MT19937(777 + b))
    def tnorm(sig):
        while True:
            v = rng.normal(0.0, sig)
            if abs(v) <= 3 * sig:
                return v
    state = np.array(cfg['srb_init'], float)
    state[3] += tnorm(2.5); state[4] += tnorm(2.5)
    state[12] += rng.normal(0.0, 0.2)
    ee = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)      # test/simulation_mpc.cpp:104-108
    return state, ee


def _cpu_rti_run(cfg, inst, steps=30):
    """one Config-B instance on the oracle: 10 cold-start solves (untimed), then `steps` timed RTI steps; returns seconds"""
    from oracle_py import OracleMPC
    dt = cfg['integrator_dt']
    s0, ee = config_b_instance(cfg, inst)
    o = OracleMPC(cfg)
    o.set_warmstart(s0)
    o.initial_run(s0, ee)
    state = s0
    t0 = time.perf_counter()
    for i in range(steps):
        t = i * dt
        eel = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        o.rti(state, t, eel)
        state = o.states()[1]
    return time.perf_counter() - t0


def cpu_baseline(cfg, seconds_budget=15.0):
    """The oracle (CPU restatement of the reference algorithm, oracle/) timed on this box's host cores: a bounded sample
    of the SAME workload -- instances 0.. of Config B, 10 cold-start solves each (untimed) then 30 timed RTI steps --
    single thread (`value`, `cores` = 1) and, as SURVEY.md 8(d) asks, over instances on all host cores of the box
    (`all_cores`: one oracle object per thread; the C++ solves release the GIL).  Reported beside the GPU number; it is
    not the target."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_py import build_oracle
    build_oracle()
    done, el, inst = 0, 0.0, 0
    while el < seconds_budget and inst < 64:
        el += _cpu_rti_run(cfg, inst)
        done += 30
        inst += 1
    out = {'value': done / el, 'unit': 'it/s', 'cores': 1, 'kind': 'port',
           'sample': '%d Config-B instances x 30 RTI steps after 10 cold-start solves each, oracle/ (C++ -O2, 1 thread)' % inst}
    try:
        ncore = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))     # 16 = the CPU share of a one-GPU box
        per_thread = 2

        def worker(k):
            return sum(_cpu_rti_run(cfg, 64 + k * per_thread + j) for j in range(per_thread))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(ncore) as ex:
            times = list(ex.map(worker, range(ncore)))
        wall = time.perf_counter() - t0
        out['all_cores'] = {'value': 30 * per_thread * ncore / max(times), 'cores': ncore, 'wall_s_incl_cold_starts': wall,
                            'sample': '%d threads x %d instances x 30 RTI steps (timed part of the slowest thread)' % (ncore, per_thread)}
    except Exception as e:        # the single-thread figure stands on its own
        out['all_cores'] = {'error': str(e)}
    return out


def pmc_traffic(steps):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE need two separate rocprofv3
    runs, so the figure is read from the committed summary of those runs, profiles/r01/k3_pmc_traffic.json, and scaled to the
    number of RTI steps of this launch; None if absent)"""
    path = os.path.join(ROOT, 'profiles', 'r01', 'k3_pmc_traffic.json')
    try:
        d = json.load(open(path))
        return float(d['hbm_bytes_per_launch']) * steps / float(d['steps_per_launch'])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='instances per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['B', 'D'], default='B',
                    help='B (default, the metric\'s configuration): 256 instances/GPU, N=20; D: 512 instances/GPU, N=50, push distribution')
    ap.add_argument('--closed-loop-steps', type=int, default=20,
                    help='extra, separately timed segment: closed-loop rollouts against the SRBM plant with pushes (0 = skip)')
    ap.add_argument('--gait-steps', type=int, default=30,
                    help='extra, separately timed segment: controller loop with the bilevel (gait) step every 5th iteration (0 = skip)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from srbm_loader import host

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    cfg = host.load_config('a1_configuration' if args.workload == 'B' else 'a1_config_distr_rejection')
    B = args.batch if args.workload == 'B' else (512 if args.batch == BATCH_PER_GPU else args.batch)
    make_instance = config_b_instance if args.workload == 'B' else config_d_instance
    lo, hi = shard_range(B * world, rank, world)            # this rank's instances of the global batch
    states, ees = zip(*[make_instance(cfg, b) for b in range(lo, hi)])
    states, ees = np.array(states), np.array(ees).reshape(hi - lo, 12)

    mpc = host.BatchMPC(cfg, hi - lo, device=local_rank)
    mpc.set_state_trajectory_warm_start(states)
    mpc.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
    mpc.create_initial_run(states, ees)                       # 10 cold-start solves (set-up, untimed)
    mpc.rti_advance(0, args.warmup)
    mpc.synchronize()
    rec = torch.zeros((hi - lo, RESULT_LD), dtype=torch.float64, device='cuda')

    it0, fl0 = mpc.work_counters()
    mpc.enable_kernel_timing(4)          # the K timed steps are ONE launch of the fused RTI kernel
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mpc.rti_advance(args.warmup, args.steps)                  # K device-resident RTI steps
    mpc.pack_results_dev(rec.data_ptr(), RESULT_LD)
    mpc.synchronize()
    allrec = gather_records(rec, world)                        # RCCL all-gather of the solved trajectories
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    k3_ms, k3_launches = mpc.kernel_timing()
    it1, fl1 = mpc.work_counters()
    st, err = mpc.status()

    # ---- second segment (Config C): the same batch continues with the gait step every 5th iteration ----
    gait_stats = None
    if args.gait_steps > 0:
        FREQ = 5
        gait = host.BatchGaitOptimizer(mpc)
        first = args.warmup + args.steps
        first += (-first) % FREQ + 1                 # start right after a multiple of FREQ: every block of 5 = 3 RTI + (RTI + GaitOpt) + LineSearch
        n_ls = sum(1 for r in range(first, first + args.gait_steps) if r % FREQ == 0)
        n_go = sum(1 for r in range(first, first + args.gait_steps) if (r + 1) % FREQ == 0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gait.rti_advance(first, args.gait_steps, FREQ)
        mpc.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el_g = max_over_ranks(time.perf_counter() - tg)
        stg, errg = mpc.status()
        n_all = B * world
        solves = n_all * ((args.gait_steps - n_ls) + 10 * n_ls)      # a line search is 10 RTI solves per instance
        gait_stats = {'workload': 'Config C protocol on the same batch: gait step every 5th iteration (gradient + LP, then 10-candidate line search)',
                      'steps': args.gait_steps, 'gait_opt_steps': n_go, 'line_search_steps': n_ls,
                      'rti_solves_per_s_incl_line_search': solves / el_g, 'gait_steps_per_s': n_all * n_ls / el_g,
                      'ms_per_step': 1e3 * el_g / args.gait_steps,
                      'statuses_after': {int(k): int(v) for k, v in zip(*np.unique(stg, return_counts=True))}, 'err_bits': int(np.bitwise_or.reduce(errg))}
    # ---- third segment: closed-loop rollouts, plant = SRBM dynamics under the current trajectory + one push per instance ----
    cl_stats = None
    if args.closed_loop_steps > 0:
        SUB = 10
        cl = host.BatchMPC(cfg, hi - lo, device=local_rank)
        cl.set_state_trajectory_warm_start(states)
        cl.create_initial_run(states, ees)
        cl.plant_set_state(states)
        rng = np.random.default_rng(777 + lo)
        imp = np.zeros((hi - lo, 6))
        imp[:, 0:2] = np.clip(rng.normal(0.0, 2.5, (hi - lo, 2)), -7.5, 7.5)       # lin-mom xy ~ N(0, 2.5^2), truncated at 3 sigma (Config D)
        imp[:, 5] = rng.normal(0.0, 0.2, hi - lo)                                   # yaw-rate ang-mom ~ N(0, 0.2^2)
        cl.plant_set_push(np.full(hi - lo, 2.5 * cfg['integrator_dt']), imp)
        cl.closed_loop_advance(0, 2, SUB, True)
        cl.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        cl.closed_loop_advance(2, args.closed_loop_steps, SUB, True)
        cl.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el_c = max_over_ranks(time.perf_counter() - tc)
        stc, errc = cl.status()
        cl_stats = {'workload': 'closed loop on a fresh copy of the batch: plant = SRBM dynamics (explicit Euler, %d sub-steps per step) under the current '
                                'trajectory, one push per instance at t = 2.5 dt (Config D distribution)' % SUB,
                    'steps': args.closed_loop_steps, 'rti_iterations_per_s': B * world * args.closed_loop_steps / el_c,
                    'ms_per_step': 1e3 * el_c / args.closed_loop_steps,
                    'statuses_after': {int(k): int(v) for k, v in zip(*np.unique(stc, return_counts=True))}, 'err_bits': int(np.bitwise_or.reduce(errc)),
                    'plant_finite': bool(np.all(np.isfinite(cl.plant_state())))}
        del cl
    ok = bool(np.all(err == 0) and np.all((st == 0) | (st == 1) | (st == 2)))
    n_inst = B * world
    value = n_inst * args.steps / elapsed

    if rank == 0:
        status_all = allrec[:, 0].cpu().numpy()
        k3_avg_s = (k3_ms / max(1, k3_launches)) * 1e-3
        flops_per_launch = (fl1 - fl0) / max(1, k3_launches)
        achieved = flops_per_launch / k3_avg_s / 1e12 if k3_avg_s > 0 else 0.0
        out = {
            'metric': 'MPC RTI iterations/sec (batched A1 SRBM, N=%d)' % cfg['num_nodes'],
            'value': value, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('Config B: %d A1 SRBM MPC instances per GPU, N=20, dt=0.05, a1_configuration.yaml values, '
                                    '10 cold-start solves then open-loop RTI steps (state := node 1)' % B) if args.workload == 'B' else
                                   ('Config D: %d A1 SRBM MPC instances per GPU, N=50, dt=0.02, a1_config_distr_rejection.yaml values, push '
                                    'distribution on the initial momentum, 10 cold-start solves then open-loop RTI steps' % B),
                       'batch_per_gpu': B, 'global_batch': n_inst, 'num_nodes': cfg['num_nodes'], 'parallelism': 'instances sharded x%d' % world,
                       'all_solved': ok, 'statuses': {int(k): int(v) for k, v in zip(*np.unique(status_all, return_counts=True))},
                       'mean_ipm_iterations': (it1 - it0) / max(1, (hi - lo) * args.steps)},
            'roofline': {'bound': 'mfma', 'kernel': 'srbm_rti_fused' if cfg['num_nodes'] <= 22 else 'srbm_rti_fused_long', 'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': pmc_traffic(args.steps) if args.workload == 'B' else None,
                         'avg_launch_ms': k3_avg_s * 1e3, 'algorithmic_flops_per_launch': flops_per_launch},
        }
        if gait_stats is not None:
            out['gait'] = gait_stats
        if cl_stats is not None:
            out['closed_loop'] = cl_stats
        if world == 1 and not args.no_cpu_baseline and args.workload == 'B':
            out['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
