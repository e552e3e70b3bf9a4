#!/usr/bin/env python3
"""Headline benchmark: MPC RTI iterations / second, batched A1 SRBM, N=20 (BASELINE.json `metric`, config[1]).

Workload (Config B): 256 independent MPC instances PER GPU, horizon N=20, dt=0.05,
apps/a1_configuration.yaml values, synthetic initial states (seeds 20240112+b, numpy MT19937).
Protocol: 10 cold-start solves per instance (MPC::CreateInitialRun, untimed set-up), W warm-up RTI steps, then the timed
region: K RTI steps with t_i = i*dt and state := node 1 of the previous trajectory (test/gait_opt_playground.cpp:113-126),
device-resident (inputs are in HBM when the region starts), closed by the all-gather of the result records.  The region is
repeated `--repeats` times back to back (the protocol simply continues): with the defaults that is BASELINE.md section 3.3's
protocol, 10 cold starts then 100 CONSECUTIVE timed steps (5 regions of 20), and `value` = instances x (steps x repeats) /
SUM of the regions -- the transient after the cold start included (the median region and the settled rate stand beside it,
named as such).  One "step" = one RTI iteration of every instance of the batch (shift -> assemble -> condense -> QP solve ->
line search -> trajectory update).

Solver mode of the headline (`--mode`, default `lower_start`): EVERY solve ends by the reference's termination test (Clarabel's
gap 1e-15 / feasibility 1e-10, clarabel_interface.cpp:165-175); in the K-step launches a solve is first attempted from the
linearisation point (srbm_set_solver_step_rule(0, 0.1)) and repeated from Clarabel's starting point when that fails.
`reference_criterion` (library default (0, 0): Clarabel's start for every solve) and `step_rule_mode` (tol_step 1e-5: a LOOSER
termination than the reference's, reported as a named secondary object, never as `value`) are the same protocol on fresh batches.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); instances are sharded over ranks (weak scaling: 256
per GPU) with no data-path collective; one all-gather of the per-instance result records (primal, dual, contact times)
closes every timed region -- by default through the C-ABI (srbm_allgather_results: in-place ncclAllGather on the batch's stream,
communicator from srbm_rccl_comm_init_rank; `--collective torch` uses torch.distributed's all_gather_into_tensor instead).  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment spawns the N ranks
itself (fresh child processes, the parent never touches a GPU); under torchrun the ranks are already there.  Rank 0 prints
ONE JSON line.  `--dry-run` walks the same launcher / sharding / gather path on CPU (gloo) without any HIP call.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from srbm_loader import workloads                      # (numpy only: the launcher parent never imports torch or touches a GPU)
from srbm_loader.workloads import shard_range, config_b_instance, config_c_instance, config_d_instance

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (public spec; the CDNA guide lists no fp64 row)
MFMA_FLOP = 2048.0           # one v_mfma_f64_16x16x4_f64: 16 x 16 x 4 multiply-adds
BATCH_PER_GPU = 256
PROFILE_ROUND = 'r05'


# ---------- helpers shared with the CPU (gloo) tests of the sharding logic ----------
def gather_records(rec, world):
    """all-gather of fixed-size per-instance result records (rows) over the ranks"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def max_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------- CPU baseline: the oracle's C++ driver, built and run on THIS box ----------
def write_cpu_baseline_input(cfg, n_inst, path):
    v = [cfg['num_nodes'], cfg['integrator_dt'], cfg['friction_coef'], cfg['force_bound'], cfg['swing_height'], cfg['foot_offset'],
         cfg['ee_box_size'][0], cfg['ee_box_size'][1], cfg['force_cost'], cfg['mass']]
    v += list(np.asarray(cfg['Ir'], float).reshape(-1)) + list(np.asarray(cfg['hip_xy'], float).reshape(-1))
    v += [float(x) for x in cfg['Q_srbd_diag']] + [float(x) for x in cfg['srb_target']]
    v.append(n_inst)
    for b in range(n_inst):
        s, ee = config_b_instance(cfg, b)
        v += list(s) + list(ee.reshape(-1))
    np.asarray(v, dtype=np.float64).tofile(path)


def host_cpu_share():
    """host threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a one-GPU box shows the
    256 hardware threads of its host but is entitled to 16 of them)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, n_single=48, per_thread=4, steps=30):
    """The oracle (CPU restatement of the reference algorithm, oracle/) timed on this box's host cores, as BASELINE.md
    section 3.4 planned: oracle/cpu_baseline.cpp compiled HERE with -O3 -march=native -ffp-contract=off + OpenMP, a bounded
    sample of the SAME workload -- Config-B instances, 10 cold-start solves each (untimed) then 30 timed RTI steps --
    (i) on one thread (`value`, `cores` = 1) and (ii) OpenMP over instances on all host threads (`all_cores`).  Reported
    beside the GPU number; it is not the target."""
    import tempfile
    odir = os.path.join(ROOT, 'oracle')
    subprocess.check_call(['make', '-s', '-C', odir, 'cpu_baseline'])
    ncpu = host_cpu_share()
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, 'in.bin')
        write_cpu_baseline_input(cfg, 256, inp)
        env = dict(os.environ, OMP_NUM_THREADS=str(ncpu), OMP_PROC_BIND='close')
        r = subprocess.run([os.path.join(odir, '_native', 'cpu_baseline'), inp, str(n_single), str(per_thread), str(steps)],
                           capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0:
        raise RuntimeError('cpu_baseline failed: ' + r.stderr[-500:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cpu_model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu_model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    s, a = d['single'], d['all_cores']
    return {'value': s['it_per_s'], 'unit': 'it/s', 'cores': 1, 'kind': 'port',
            'sample': '%d Config-B instances x %d RTI steps after 10 cold-start solves each; oracle/cpu_baseline.cpp, g++ -O3 -march=native '
                      '-ffp-contract=off, 1 thread, %.1f s, mean %.1f IPM iterations per solve' % (s['instances'], s['steps'], s['seconds'], s['mean_ipm_iterations']),
            'all_cores': {'value': a['it_per_s'], 'cores': a['threads'], 'wall_s_incl_cold_starts': a['wall_s_incl_cold_starts'],
                          'sample': 'OpenMP, %d threads x %d instances x %d RTI steps (timed part of the slowest thread)' % (a['threads'], per_thread, a['steps'])},
            'cpu_model': cpu_model, 'nproc': ncpu, 'not_solved': s['not_solved'] + a['not_solved']}


# ---------- HBM traffic of the dominant kernel from the committed PMC summary ----------
def kernel_source_sha():
    """sha256 over the kernel sources + C-ABI header: a PMC summary is only quoted for the code it was measured on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'bilevel-gait-gen_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hiph', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(steps):
    """HBM bytes per launch of the dominant kernel from the two PMC passes (FETCH_SIZE / WRITE_SIZE need separate rocprofv3 runs,
    scripts/collect_profiles.sh), read from the committed summary profiles/<round>/pmc_summary.json and scaled to the RTI steps
    of one launch.  None unless the summary was taken on exactly these kernel sources."""
    path = os.path.join(ROOT, 'profiles', PROFILE_ROUND, 'pmc_summary.json')
    try:
        d = json.load(open(path))
        if d.get('kernel_source_sha') != kernel_source_sha():
            return None, None
        return float(d['hbm_bytes_per_step']) * steps, d
    except Exception:
        return None, None


# ---------- rank launcher ----------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start N fresh rank processes of this same script.  The parent has not
    imported torch and never touches a GPU; nothing is exec'd from a GPU-initialised process."""
    # the ranks would otherwise race on `make` when a library is missing (fresh box): build once, here (make only -- the parent never touches a GPU)
    if not args.dry_run:
        from srbm_loader import host
        host.build()
    port = free_port()
    procs = []
    for r in range(args.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC -- without it RCCL's cross-process buffer exchange fails
        # with `hipIpcGetMemHandle: invalid argument`; kept if the caller already exports it, set otherwise
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: the first rank that fails ends the run (its peers would otherwise sit in the rendezvous / a barrier until the RCCL timeout)
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                if r != 0:
                    rc = abs(r)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps run back to back: value = instances x steps x repeats / their SUM')
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='instances per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['B', 'D', 'E'], default='B',
                    help='B (default, the metric\'s configuration): 256 instances/GPU, N=20; D: 512 instances/GPU, N=50, push distribution; '
                         'E: 128 instances/GPU, N=40 (2 s horizon, 232 spline variables) on the LARGE-capacity build')
    ap.add_argument('--closed-loop-steps', type=int, default=20,
                    help='extra, separately timed segment: closed-loop rollouts against the SRBM plant with pushes (0 = skip)')
    ap.add_argument('--gait-steps', type=int, default=30,
                    help='extra, separately timed segment (Config C): controller loop with the bilevel (gait) step every 5th iteration (0 = skip)')
    ap.add_argument('--wbc-ticks', type=int, default=20,
                    help='fourth segment: 1 kHz control ticks (targets from the trajectory by IK + whole-body QP) of the batch; 0 skips it')
    ap.add_argument('--steady-from', type=int, default=125,
                    help='first RTI step of the extra `steady_state` regions (the protocol of the headline continued on the same batch past the '
                         'transient that follows the cold start; 0 = skip).  The headline `value` is not affected')
    ap.add_argument('--mode', choices=['lower_start', 'reference', 'step_rule'], default='lower_start',
                    help='solver mode of the headline: lower_start (default) = every solve ends by the reference\'s gap criterion, attempted first from the '
                         'linearisation point (0, 0.1); reference = the library default (0, 0); step_rule = tol_step 1e-5 + lower start (looser termination)')
    ap.add_argument('--no-reference-criterion', '--no-other-modes', dest='no_other_modes', action='store_true',
                    help='skip the runs of the Config-B protocol in the two other solver modes')
    ap.add_argument('--collective', choices=['cabi', 'torch'], default='cabi',
                    help='all-gather of the result records: cabi (default) = srbm_allgather_results (RCCL through the C-ABI), torch = torch.distributed')
    ap.add_argument('--extra-workloads', type=int, default=1, help='1 (default): short Config D and Config E runs quoted as config_d / config_e objects (workload B only)')
    ap.add_argument('--n1-baseline', type=float, default=float(os.environ.get('SRBM_BENCH_N1_BASELINE', '0') or 0),
                    help='it/s of the 1-GPU run of the same command: the line then carries weak_scaling_efficiency = value / (n_gpus * this)')
    ap.add_argument('--dry-run', action='store_true', help='CPU rehearsal of the launcher / sharding / gather path (gloo, no HIP call)')
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.repeats < 1:
        raise SystemExit('bench.py: --gpus, --steps and --repeats must be positive')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a wrong n_gpus\n' % (args.gpus, world))
        sys.exit(3)

    import torch
    import torch.distributed as dist
    from srbm_loader import host

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.workload == 'E':       # SURVEY.md 8d Config E: the SRBM path at N = 40, dt = 0.05 (the reference's centroidal MPC is dead code)
        cfg = host.load_config('a1_configuration', num_nodes=40)
    else:
        cfg = host.load_config('a1_configuration' if args.workload == 'B' else 'a1_config_distr_rejection')
    B = args.batch if args.workload == 'B' else ((512 if args.workload == 'D' else 128) if args.batch == BATCH_PER_GPU else args.batch)
    make_instance = config_d_instance if args.workload == 'D' else config_b_instance
    lo, hi = shard_range(B * world, rank, world)            # this rank's instances of the global batch
    n_inst = B * world

    if args.dry_run:
        # same sharding and the same collective on CPU tensors; the records carry the global instance index instead of a solution
        if world > 1:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        ld = 16
        rec = torch.zeros((hi - lo, ld), dtype=torch.float64)
        rec[:, 0] = 0.0
        rec[:, 1] = torch.arange(lo, hi, dtype=torch.float64)
        allrec = gather_records(rec, world)
        ok = bool(torch.equal(allrec[:, 1], torch.arange(n_inst, dtype=torch.float64)))
        el = max_over_ranks(1.0)
        if rank == 0:
            print(json.dumps({'metric': 'MPC RTI iterations/sec (batched A1 SRBM, N=%d)' % cfg['num_nodes'], 'value': 0.0, 'unit': 'it/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'scaling': 'weak',
                              'collective_world_size': dist.get_world_size() if world > 1 else 1,
                              'config': {'batch_per_gpu': B, 'global_batch': n_inst, 'records_gathered': int(allrec.shape[0]), 'gather_ok': ok,
                                         'shard_bounds': [list(shard_range(n_inst, r_, world)) for r_ in range(world)]}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0 if ok and el == 1.0 else 4)

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    # SRBM_BENCH_FORCE_DIST=1: initialise the RCCL process group and run every collective of the N > 1 path with a world of one rank
    # (rehearsal of that path on a one-GPU box)
    DIST = world > 1 or os.environ.get('SRBM_BENCH_FORCE_DIST') == '1'
    if DIST:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(free_port())      # (one-rank rehearsal: no peer needs to know it)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    # The all-gather that closes a timed region.  Default: through the C-ABI -- the library packs this rank's records into their slot of the
    # gathered array and runs one in-place ncclAllGather on the batch's stream (include/srbm_rti.h: srbm_allgather_results); the communicator is
    # made by the library's helpers, its unique id travels over the process group that is there for the barriers.  A batch that cannot make one
    # says so on stderr and the run continues on torch.distributed's collective, named in the JSON line.
    coll = {'kind': 'none (one rank, no process group)' if not DIST else args.collective, 'comm': None, 'note': None}

    def make_comm(m):
        """ONE ncclComm_t per rank for all batches of the run (a communicator belongs to the device, not to a batch; created outside every timed
        region through the first batch that gathers)"""
        idb = [m.rccl_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(idb, src=0)
        return m.rccl_comm_init_rank(world, rank, idb[0])

    def gather_dev(m, rec_all, rec_mine):
        """all ranks' records into rec_all [world * batch][ld] (returns it); rec_mine: this rank's slot (a view) for the torch path"""
        if not DIST:
            m.pack_results_dev(rec_all.data_ptr(), rec_all.shape[1])
            m.synchronize()
            return rec_all
        if coll['kind'] == 'cabi':
            if coll['comm'] is None:
                try:
                    coll['comm'] = make_comm(m)
                except Exception as e:          # loud, and recorded in the line: never a silent switch
                    sys.stderr.write('bench.py: C-ABI RCCL communicator failed (%s); continuing on torch.distributed all_gather\n' % e)
                    coll['kind'] = 'torch'; coll['note'] = 'srbm_rccl_comm_init_rank failed: %s' % e
            if coll['kind'] == 'cabi':
                m.allgather_results(coll['comm'], rec_all.data_ptr())
                m.synchronize()
                return rec_all
        m.pack_results_dev(rec_mine.data_ptr(), rec_mine.shape[1])
        m.synchronize()
        dist.all_gather_into_tensor(rec_all, rec_mine)
        return rec_all

    def solved_quality(acc):
        """[error bits, solves, not solved, max-iter] of ALL timed solves (sticky accumulators), reduced over the ranks"""
        q = np.array([float(np.bitwise_or.reduce(acc[:, 0])), float(acc[:, 1].sum()), float(acc[:, 2].sum()), float(acc[:, 3].sum())])
        if DIST:
            tq = torch.tensor(q, dtype=torch.float64, device='cuda')
            gl = [torch.zeros_like(tq) for _ in range(world)]
            dist.all_gather(gl, tq)
            allq = torch.stack(gl).cpu().numpy()
            q = np.array([float(np.bitwise_or.reduce(allq[:, 0].astype(np.int64))), allq[:, 1].sum(), allq[:, 2].sum(), allq[:, 3].sum()])
        return q

    def set_mode(m, mode):
        if mode == 'lower_start':
            m.enable_lower_start()                            # srbm_set_solver_step_rule(0, 0.1): reference's termination test, lower starting point
        elif mode == 'step_rule':
            m.enable_fast_termination()                       # srbm_set_solver_step_rule(1e-5, 0.1)
        else:
            m.set_solver_step_rule(0.0, 0.0)                  # the library default = the reference's solver settings

    def run_protocol(cfg_w, make_inst, per_gpu, large, mode, warmup, steps, repeats):
        """The timed protocol on this rank's shard of `per_gpu * world` instances: 10 cold-start solves (untimed), `warmup` RTI steps, then `repeats`
        regions of `steps` device-resident RTI steps, each closed by packing the result records and the all-gather over the ranks; barrier +
        synchronize on both sides of every region, max over ranks.  mode: lower_start | reference | step_rule (set_mode)."""
        lo_, hi_ = shard_range(per_gpu * world, rank, world)
        st_, ee_ = zip(*[make_inst(cfg_w, b) for b in range(lo_, hi_)])
        st_, ee_ = np.array(st_), np.array(ee_).reshape(hi_ - lo_, 12)
        m = host.BatchMPC(cfg_w, hi_ - lo_, device=local_rank, large=large)
        m.set_state_trajectory_warm_start(st_)
        m.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)     # ClarabelInterface's settings (clarabel_interface.cpp:165-175)
        set_mode(m, mode)
        m.create_initial_run(st_, ee_)                        # 10 cold-start solves (set-up, untimed)
        m.rti_advance(0, warmup)
        m.synchronize()
        ld = m.result_record_doubles()                        # status..., primal, dual, contact times (SURVEY.md 8e record)
        nb_ = hi_ - lo_
        rec_all = torch.zeros((world * nb_, ld), dtype=torch.float64, device='cuda')
        rec_mine = rec_all[rank * nb_:(rank + 1) * nb_]
        if DIST:                                              # first collective outside the timed region (communicator set-up)
            gather_dev(m, rec_all, rec_mine)
            torch.cuda.synchronize()
        m.clear_status_accumulators()
        prev = m.work_counters() + (m.executed_mfma(),)
        it_first = prev[0]
        m.enable_kernel_timing(repeats + 2)                   # every timed region is ONE launch of the fused RTI kernel
        reg_s, reg_local, reg_work = [], [], []               # max over ranks; this rank's own; (IPM iterations, algorithmic flops, executed MFMA) per region
        first = warmup
        for rep in range(repeats):
            if DIST:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.rti_advance(first, steps)                       # K device-resident RTI steps
            gather_dev(m, rec_all, rec_mine)                  # RCCL all-gather of the solved trajectories (primal + dual + schedule)
            torch.cuda.synchronize()
            if DIST:
                dist.barrier()
            loc = time.perf_counter() - t0
            reg_local.append(loc)
            reg_s.append(max_over_ranks(loc))
            ctr = m.work_counters() + (m.executed_mfma(),)    # (outside the timed region)
            reg_work.append(tuple(x - y for x, y in zip(ctr, prev)))
            prev = ctr
            first += steps
        med_ = int(np.argsort(reg_s)[len(reg_s) // 2])
        acc = m.status_accumulated()
        return dict(mpc=m, lo=lo_, hi=hi_, n_inst=per_gpu * world, region_s=reg_s, region_local_s=reg_local, region_work=reg_work, med=med_,
                    elapsed=float(np.sum(reg_s)), median_region=float(np.median(reg_s)), launch_ms=m.kernel_timings(repeats + 2), acc=acc, q=solved_quality(acc),
                    counters=m.solver_counters(), allrec=rec_all, ld=ld, states=st_, ees=ee_, mode=mode,
                    mean_iters=(prev[0] - it_first) / max(1, (hi_ - lo_) * steps * repeats))

    def summary(r, steps, label):
        """short object for the extra workloads / criteria quoted beside the headline"""
        q_ = r['q']
        el, nrep = r['elapsed'], len(r['region_s'])
        m = r['mpc']
        ts, mu = m.solver_step_rule()
        return {'workload': label, 'mode': r['mode'], 'value': r['n_inst'] * steps * nrep / el, 'unit': 'it/s', 'ms_per_step': 1e3 * el / (steps * nrep), 'steps': steps, 'repeats': nrep,
                'statistic': 'instances x steps x repeats / SUM of the regions (consecutive steps after the cold start)',
                'value_from_median_region': r['n_inst'] * steps / r['median_region'],
                'region_ms': [1e3 * v for v in r['region_s']], 'batch_per_gpu': r['hi'] - r['lo'], 'global_batch': r['n_inst'], 'num_nodes': m.N,
                'large_build': bool(m.large), 'records_gathered': int(r['allrec'].shape[0]),
                'mean_ipm_iterations': r['mean_iters'], 'timed_solves': int(q_[1]), 'not_solved_in_timed_solves': int(q_[2]),
                'max_iter_in_timed_solves': int(q_[3]), 'err_bits_all_timed_steps': int(q_[0]), 'all_solved': bool(q_[2] == 0 and q_[0] == 0),
                'instances_with_a_solve_not_solved_rank0': [int(r['lo'] + b) for b in np.nonzero(r['acc'][:, 2])[0][:64]],
                'solver': {'tol_gap': 1e-15, 'tol_feas': 1e-10, 'tol_step': ts, 'start_mu': mu, 'solves_ended_by_step_rule': r['counters']['step_rule'],
                           'lower_start_attempts': r['counters']['low_tried'], 'attempts_repeated_from_standard_start': r['counters']['low_failed']}}

    MODE = args.mode
    MODE_TEXT = {'lower_start': 'every solve ENDS by the reference criterion (gap 1e-15, feasibility 1e-10) and is first attempted from the linearisation point with '
                                'centred multipliers (srbm_set_solver_step_rule(0, 0.1)): same termination test as the reference, other starting point',
                 'reference': 'every solve to the reference criterion from Clarabel\'s starting point (srbm_set_solver_step_rule(0, 0): the library default, what a '
                              'caller of the mpc:: facade gets)',
                 'step_rule': 'solves end by the STEP RULE (tol_step 1e-5: the affine Newton step bounds the distance to the minimiser) -- a looser termination '
                              'than the reference\'s gap 1e-15 -- with the lower starting point (srbm_set_solver_step_rule(1e-5, 0.1))'}
    main = run_protocol(cfg, make_instance, B, args.workload == 'E', MODE, args.warmup, args.steps, args.repeats)
    mpc, states, ees = main['mpc'], main['states'], main['ees']
    region_s, region_work, med, elapsed = main['region_s'], main['region_work'], main['med'], main['elapsed']
    k3_each = main['launch_ms']; k3_launches = len(k3_each)
    acc_main, ctrs, allrec, LD, q = main['acc'], main['counters'], main['allrec'], main['ld'], main['q']

    # ---- the same protocol in the two OTHER solver modes, on fresh batches: `reference_criterion` = the library default (0, 0), the like-for-like
    # number beside the CPU baseline, which runs Clarabel's start to Clarabel's criterion too; `step_rule_mode` = the looser termination ----
    other = {}
    if args.workload == 'B' and not args.no_other_modes:
        for mo in ('reference', 'lower_start', 'step_rule'):
            if mo != MODE:
                rr = run_protocol(cfg, make_instance, B, False, mo, args.warmup, args.steps, args.repeats)
                other[mo] = summary(rr, args.steps, 'Config B, same protocol: ' + MODE_TEXT[mo])
                del rr
        # the same steps of the headline's mode as ONE region (one launch of steps x repeats steps, one all-gather): what the four synchronisation
        # points between the regions of `value` cost -- at each of them the batch waits for the instance whose 20 solves happened to be the longest
        if args.repeats > 1:
            rr = run_protocol(cfg, make_instance, B, False, MODE, args.warmup, args.steps * args.repeats, 1)
            other['one_region'] = summary(rr, args.steps * args.repeats, 'Config B, the %d steps of `value` as ONE timed region (one launch): ' % (args.steps * args.repeats) + MODE_TEXT[MODE])
            del rr
    # ---- BASELINE configs 4 and 5 at their per-GPU sizes, the same protocol with the same --steps / --warmup / --repeats, in the headline's mode
    # (Config D also in the step-rule mode: the number rounds 3-4 tracked) ----
    d_stats = e_stats = None
    if args.workload == 'B' and args.extra_workloads:
        cfg_d = host.load_config('a1_config_distr_rejection')
        d_label = ('Config D: 512 A1 SRBM MPC instances per GPU, N=50, dt=0.02, a1_config_distr_rejection.yaml values, push distribution on the '
                   'initial momentum; not-solved solves belong to instances whose QPs the oracle finds infeasible too')
        rd = run_protocol(cfg_d, config_d_instance, 512, False, MODE, args.warmup, args.steps, args.repeats)
        d_stats = summary(rd, args.steps, d_label)
        del rd
        if MODE != 'step_rule':
            rd = run_protocol(cfg_d, config_d_instance, 512, False, 'step_rule', args.warmup, args.steps, args.repeats)
            d_stats['step_rule_mode'] = summary(rd, args.steps, d_label)
            del rd
        cfg_e = host.load_config('a1_configuration', num_nodes=40)
        re_ = run_protocol(cfg_e, config_b_instance, 128, True, MODE, args.warmup, args.steps, args.repeats)
        e_stats = summary(re_, args.steps, 'Config E (SRBM stand-in for the dead centroidal MPC): 128 instances per GPU, N=40, dt=0.05, LARGE-capacity build')
        del re_

    # ---- second segment (Config C): N=20 / dt=0.05 with the values of apps/a1_gait_opt_config.yaml, gait step every 5th iteration.
    # No lower-start attempts in this protocol (one-step launches).  In the headline's default mode every solve of the segment runs to the gap
    # criterion; `step_rule_mode` beside it = plain steps and line-search candidates end by the step rule (the differentiated solve always runs to the
    # gap criterion: srbm_gait_rti_advance does that by itself) ----
    def gait_segment(step_rule):
        FREQ = 5
        cfg_c = host.load_config('a1_gait_opt_config', num_nodes=20, integrator_dt=0.05)
        sc, ec = zip(*[config_c_instance(cfg_c, b) for b in range(lo, hi)])
        sc, ec = np.array(sc), np.array(ec).reshape(hi - lo, 12)
        gm = host.BatchMPC(cfg_c, hi - lo, device=local_rank)
        gm.set_state_trajectory_warm_start(sc)
        gm.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
        if step_rule:
            gm.enable_fast_termination()
        gm.create_initial_run(sc, ec)
        gait = host.BatchGaitOptimizer(gm)
        gait.rti_advance(0, 6, FREQ)                 # run_num 0..5: includes one gradient + LP (run 4) and one line search (run 5)
        gm.synchronize()
        gm.clear_status_accumulators()
        firstg = 6
        n_ls = sum(1 for r in range(firstg, firstg + args.gait_steps) if r % FREQ == 0)
        n_go = sum(1 for r in range(firstg, firstg + args.gait_steps) if (r + 1) % FREQ == 0)
        if DIST:
            dist.barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gait.rti_advance(firstg, args.gait_steps, FREQ)
        gm.synchronize()
        torch.cuda.synchronize()
        if DIST:
            dist.barrier()
        el_g = max_over_ranks(time.perf_counter() - tg)
        stg, _ = gm.status()
        accg = gm.status_accumulated()
        lp_st, _ = gait.lp_result()
        solves = n_inst * ((args.gait_steps - n_ls) + 10 * n_ls)      # a line search is 10 RTI solves per instance
        out_g = {'workload': 'Config C: %d instances per GPU, N=20, dt=0.05, a1_gait_opt_config.yaml values (mu 0.6, force bound 200, target x=y=1); '
                             'controller protocol with the gait step every 5th iteration (gradient + LP, then 10-candidate line search)' % B,
                 'solver_tol_step': gm.solver_step_rule()[0],
                 'steps': args.gait_steps, 'gait_opt_steps': n_go, 'line_search_steps': n_ls,
                 'rti_solves_per_s_incl_line_search': solves / el_g, 'gait_steps_per_s': n_inst * n_ls / el_g,
                 'ms_per_step': 1e3 * el_g / args.gait_steps,
                 'statuses_after': {int(k): int(v) for k, v in zip(*np.unique(stg, return_counts=True))},
                 'err_bits_all_steps': int(np.bitwise_or.reduce(accg[:, 0])), 'not_solved_all_steps': int(accg[:, 2].sum()),
                 'lp_status_last': {int(k): int(v) for k, v in zip(*np.unique(lp_st, return_counts=True))}}
        del gait, gm
        return out_g

    gait_stats = None
    if args.gait_steps > 0 and args.workload == 'B':
        gait_stats = gait_segment(MODE == 'step_rule')
        if MODE != 'step_rule':
            gait_stats['step_rule_mode'] = gait_segment(True)
    # ---- third segment: closed-loop rollouts, plant = SRBM dynamics under the current trajectory + one push per instance ----
    cl_stats = None
    if args.closed_loop_steps > 0:
        SUB = 10
        cl = host.BatchMPC(cfg, hi - lo, device=local_rank, large=mpc.large)
        cl.set_state_trajectory_warm_start(states)
        if MODE == 'step_rule':
            cl.enable_fast_termination()     # (step rule only: under a plant the library ignores start_mu)
        cl.create_initial_run(states, ees)
        cl.plant_set_state(states)
        rng = np.random.default_rng(777 + lo)
        imp = np.zeros((hi - lo, 6))
        imp[:, 0:2] = np.clip(rng.normal(0.0, 2.5, (hi - lo, 2)), -7.5, 7.5)       # lin-mom xy ~ N(0, 2.5^2), truncated at 3 sigma (Config D)
        imp[:, 5] = rng.normal(0.0, 0.2, hi - lo)                                   # yaw-rate ang-mom ~ N(0, 0.2^2)
        cl.plant_set_push(np.full(hi - lo, 2.5 * cfg['integrator_dt']), imp)
        cl.closed_loop_advance(0, 2, SUB, True)
        cl.synchronize()
        cl.clear_status_accumulators()
        if DIST:
            dist.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        cl.closed_loop_advance(2, args.closed_loop_steps, SUB, True)
        cl.synchronize()
        torch.cuda.synchronize()
        if DIST:
            dist.barrier()
        el_c = max_over_ranks(time.perf_counter() - tc)
        stc, _ = cl.status()
        accc = cl.status_accumulated()
        cl_stats = {'workload': 'closed loop on a fresh copy of the batch: plant = SRBM dynamics (explicit Euler, %d sub-steps per step) under the current '
                                'trajectory, one push per instance at t = 2.5 dt (Config D distribution)' % SUB,
                    'solver_tol_step': cl.solver_step_rule()[0],
                    'steps': args.closed_loop_steps, 'rti_iterations_per_s': n_inst * args.closed_loop_steps / el_c,
                    'ms_per_step': 1e3 * el_c / args.closed_loop_steps,
                    'statuses_after': {int(k): int(v) for k, v in zip(*np.unique(stc, return_counts=True))},
                    'err_bits_all_steps': int(np.bitwise_or.reduce(accc[:, 0])), 'not_solved_all_steps': int(accc[:, 2].sum()),
                    'plant_finite': bool(np.all(np.isfinite(cl.plant_state())))}
        del cl
    # ---- fourth segment: the 1 kHz step downstream of the MPC for the whole batch: MPCController::GetTargetsFromTraj (two IK
    # solves per instance) then QPControl::ComputeControlAction (rigid-body dynamics + the whole-body QP), through the host-pointer entries ----
    wbc_stats = None
    if args.wbc_ticks > 0 and args.workload == 'B' and 'init_config' in cfg:
        nb = hi - lo
        t0w = mpc.get_trajectory(0, 1)[0].init_time
        q_des = np.tile(np.array(cfg['init_config'], float), (nb, 1))
        q_des, v_des, f_des, st_t = mpc.get_targets_from_traj(t0w, q_des)           # warm-up tick (also the IK guess of the first timed one)
        rngw = np.random.default_rng(4242 + lo)
        bad_t = bad_q = 0; qp_it = []
        torch.cuda.synchronize()
        tw = time.perf_counter(); t_targets = 0.0; t_qp = 0.0
        for k in range(args.wbc_ticks):
            tk = t0w + 1e-3 * (k + 1)
            ta = time.perf_counter()
            q_des, v_des, f_des, st_t = mpc.get_targets_from_traj(tk, q_des)
            t_targets += time.perf_counter() - ta
            _, _, con = mpc.eval_trajectory(tk)
            # force_target_: 3 per foot in contact, stacked in foot order (stable sort of the feet by "not in contact")
            order = np.argsort(con == 0, axis=1, kind='stable')
            fd = (np.take_along_axis(f_des, order[:, :, None], axis=1) * (np.take_along_axis(con, order, axis=1) > 0)[:, :, None]).reshape(nb, 12)
            q_meas = q_des.copy(); q_meas[:, 7:] += rngw.normal(size=(nb, 12)) * 0.01  # "measured" state: the target with a tracking error
            v_meas = v_des + rngw.normal(size=(nb, 18)) * 0.01
            ta = time.perf_counter()
            ctl, sol, st_q, it_q = mpc.qp_control(q_meas, v_meas, con, q_des, v_des, fd)
            t_qp += time.perf_counter() - ta
            bad_t += int((st_t != 0).sum()); bad_q += int((st_q > 1).sum()); qp_it.append(it_q.mean())
        el_w = max_over_ranks(time.perf_counter() - tw)
        # ... and the same ticks with everything RESIDENT on the device: the device-pointer entries on the batch's own stream (torch only allocates
        # and stacks the contact forces, on that stream), no copy and no synchronisation inside a tick -- what a batched controller would run
        ext = torch.cuda.ExternalStream(mpc.stream())
        with torch.cuda.stream(ext):
            f64 = dict(dtype=torch.float64, device='cuda')
            t_d = torch.zeros(nb, **f64); q_d = torch.tensor(q_des, **f64); v_d = torch.zeros((nb, 18), **f64); f_d = torch.zeros((nb, 4, 3), **f64)
            st_d = torch.zeros(nb, dtype=torch.int32, device='cuda'); ef_d = torch.zeros((nb, 12), **f64); ep_d = torch.zeros((nb, 12), **f64)
            con_d = torch.zeros((nb, 4), dtype=torch.int32, device='cuda'); ctl_d = torch.zeros((nb, 36), **f64); sol_d = torch.zeros((nb, 30), **f64)
            stq_d = torch.zeros(nb, dtype=torch.int32, device='cuda')
            gen = torch.Generator(device='cuda'); gen.manual_seed(4242 + lo)
            bad_acc = torch.zeros(2, dtype=torch.int64, device='cuda')

            def dev_tick(tk):
                t_d.fill_(tk)
                mpc.get_targets_from_traj_dev(t_d.data_ptr(), q_d.data_ptr(), v_d.data_ptr(), f_d.data_ptr(), st_d.data_ptr())
                mpc.eval_trajectory_dev(t_d.data_ptr(), ef_d.data_ptr(), ep_d.data_ptr(), con_d.data_ptr())
                order = torch.argsort((con_d == 0).to(torch.uint8), dim=1, stable=True)
                fd_d = (torch.take_along_dim(f_d, order[:, :, None], dim=1) * (torch.take_along_dim(con_d, order, dim=1) > 0)[:, :, None]).reshape(nb, 12).contiguous()
                qm_d = q_d.clone(); qm_d[:, 7:] += 0.01 * torch.randn((nb, 12), generator=gen, **f64)
                vm_d = v_d + 0.01 * torch.randn((nb, 18), generator=gen, **f64)
                mpc.qp_control_dev(qm_d.data_ptr(), vm_d.data_ptr(), con_d.data_ptr(), q_d.data_ptr(), v_d.data_ptr(), fd_d.data_ptr(), ctl_d.data_ptr(), sol_d.data_ptr(),
                                   stq_d.data_ptr())
                bad_acc[0] += (st_d != 0).sum(); bad_acc[1] += ((stq_d & 255) > 1).sum()
                return fd_d, qm_d, vm_d                      # (kept alive until the stream has used them)

            keep = dev_tick(t0w + 1e-3 * (args.wbc_ticks + 1))      # warm-up
            mpc.synchronize(); bad_acc.zero_()
            td0 = time.perf_counter()
            for k in range(args.wbc_ticks): keep = dev_tick(t0w + 1e-3 * (args.wbc_ticks + 2 + k))
            mpc.synchronize()
            el_d = max_over_ranks(time.perf_counter() - td0)
            bad_d = bad_acc.cpu().numpy(); ctl_fin = bool(torch.isfinite(ctl_d).all().item())
        wbc_stats = {'workload': '1 kHz control ticks downstream of the MPC for the %d instances of the batch: GetTargetsFromTraj (linear state interpolation, '
                                 'two IK solves, force splines) + QPControl::ComputeControlAction (dynamics by recursive Newton-Euler, whole-body QP), '
                                 'host-pointer entries (PCIe copies and the host-side stacking of the contact forces included)' % nb,
                     'ticks': args.wbc_ticks, 'control_actions_per_s': n_inst * args.wbc_ticks / el_w, 'ms_per_tick_of_the_batch': 1e3 * el_w / args.wbc_ticks,
                     'ms_per_tick_targets_only': 1e3 * t_targets / args.wbc_ticks, 'ms_per_tick_qp_control_only': 1e3 * t_qp / args.wbc_ticks,
                     'targets_not_ok': bad_t, 'qp_not_solved': bad_q, 'mean_qp_ipm_iterations': float(np.mean(qp_it)), 'finite': bool(np.all(np.isfinite(ctl))) and ctl_fin,
                     'device_resident': {'note': 'the same ticks through the device-pointer entries on the batch\'s stream (srbm_get_targets_from_traj_dev, srbm_eval_trajectory_dev, '
                                                 'srbm_qp_control_dev; contact forces stacked on the device): no copy, no synchronisation inside a tick',
                                         'ms_per_tick_of_the_batch': 1e3 * el_d / args.wbc_ticks, 'control_actions_per_s': n_inst * args.wbc_ticks / el_d,
                                         'targets_not_ok': int(bad_d[0]), 'qp_not_solved': int(bad_d[1])}}
    value = n_inst * args.steps * args.repeats / elapsed          # `elapsed` = SUM of the timed regions

    # ---- the headline's protocol CONTINUED on the same batch: `repeats` more regions of `steps` steps from step `--steady-from` on.  The five
    # regions of the headline cover the first ~100 steps after the cold start, where the lower-start attempts still fail more often and the
    # solves are longer; a controller that runs for seconds sees the rate below.  Reported beside `value`, never instead of it ----
    steady = None
    if args.steady_from > 0 and args.workload == 'B':
        first_s = args.warmup + args.steps * args.repeats
        if args.steady_from > first_s:
            mpc.rti_advance(first_s, args.steady_from - first_s)          # untimed
            first_s = args.steady_from
        rec_s = allrec
        rec_s_mine = rec_s[rank * (hi - lo):(rank + 1) * (hi - lo)]
        mpc.synchronize()
        reg_steady = []
        for rep in range(args.repeats):
            if DIST:
                dist.barrier()
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            mpc.rti_advance(first_s, args.steps)
            gather_dev(mpc, rec_s, rec_s_mine)
            torch.cuda.synchronize()
            if DIST:
                dist.barrier()
            reg_steady.append(max_over_ranks(time.perf_counter() - t0s))
            first_s += args.steps
        el_s = float(np.sum(reg_steady))
        st_s, err_s = mpc.status()
        steady = {'value': n_inst * args.steps * args.repeats / el_s, 'unit': 'it/s', 'ms_per_step': 1e3 * el_s / (args.steps * args.repeats), 'region_ms': [1e3 * v for v in reg_steady],
                  'first_step': first_s - args.steps * args.repeats, 'statistic': 'instances x steps x repeats / SUM of %d regions of %d steps, same bracket as the headline' % (args.repeats, args.steps),
                  'all_solved_after': bool(np.all(st_s <= 1) and np.all(err_s == 0)),
                  'note': 'the protocol of `value` continued on the same batch past the transient after the cold start (its regions are steps %d..%d); '
                          '`value` and `roofline` describe the regions of the headline only' % (args.warmup, args.warmup + args.steps * args.repeats)}

    # this rank's own region times, gathered: a straggling GPU shows up here, not only in the max
    per_rank_ms = [[1e3 * v for v in main['region_local_s']]]
    if DIST:
        tl = torch.tensor(main['region_local_s'], dtype=torch.float64, device='cuda')
        gl = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(gl, tl)
        per_rank_ms = [[1e3 * float(v) for v in t_.cpu().numpy()] for t_ in gl]
    if rank == 0:
        status_all = allrec[:, 0].cpu().numpy()
        err_all = allrec[:, 5].cpu().numpy().astype(np.int64)
        if allrec.shape[0] != n_inst:
            raise SystemExit('gathered %d records for %d instances' % (allrec.shape[0], n_inst))
        # the roofline object describes the SAME regions `value` does: all timed launches -- average launch duration (HIP events on the launch
        # stream), average flop and MFMA counts per launch
        n_l = min(k3_launches, len(region_work))
        k3_avg_s = float(np.mean(k3_each[:n_l])) * 1e-3 if n_l else 0.0
        flops_per_launch = float(np.mean([w_[1] for w_ in region_work[:n_l]])) if n_l else 0.0
        mfma_per_launch = float(np.mean([w_[2] for w_ in region_work[:n_l]])) if n_l else 0.0
        achieved = flops_per_launch / k3_avg_s / 1e12 if k3_avg_s > 0 else 0.0
        exec_tflops = mfma_per_launch * MFMA_FLOP / k3_avg_s / 1e12 if k3_avg_s > 0 else 0.0
        traffic, pmc = pmc_traffic(args.steps) if args.workload == 'B' else (None, None)
        # `bound` keeps to the schema of the measurement contract (hbm | mfma): it names the roof `frac` is priced against.  What limits the
        # kernel in practice is neither -- `limiter` says so in one word, `limiter_detail` in a sentence
        roof = {'bound': 'mfma', 'limiter': 'latency',
                'limiter_detail': 'latency of dependent chains (1 workgroup of 8 waves per CU, no HBM or matrix-pipe saturation); frac is measured against '
                                  'the fp64 matrix roof, the nearest one',
                'statistic': 'averages over the %d timed launches (one per region) that `value` is computed from' % n_l,
                'launch_ms_all': [float(v) for v in k3_each],
                'kernel': ('srbm_rti_fused' if cfg['num_nodes'] <= 22 else 'srbm_rti_fused_long') + (' (LARGE build)' if mpc.large else ''),
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP64_PEAK_TFLOPS,
                'achieved_is': 'ALGORITHMIC flops (SURVEY.md 8d formula at the sizes and IPM iteration counts executed) / measured launch time',
                'traffic': traffic, 'avg_launch_ms': k3_avg_s * 1e3, 'launches_timed': k3_launches, 'ipm_iterations_per_solve': float(np.mean([w_[0] for w_ in region_work[:n_l]])) / max(1, (hi - lo) * args.steps) if n_l else 0.0,
                'algorithmic_flops_per_launch': flops_per_launch,
                'executed_mfma_tflops': exec_tflops, 'executed_mfma_frac_of_peak': exec_tflops / FP64_PEAK_TFLOPS,
                'executed_mfma_instructions_per_launch': mfma_per_launch}
        if pmc is not None:
            roof['traffic_source'] = 'profiles/%s/pmc_summary.json (kernel sources %s)' % (PROFILE_ROUND, pmc['kernel_source_sha'])
            for k in ('mfma_busy_frac', 'lds_bank_conflict_frac', 'valu_busy_frac', 'occupancy_waves_per_cu'):
                if k in pmc:
                    roof['pmc_' + k] = pmc[k]
        out = {
            'metric': 'MPC RTI iterations/sec (batched A1 SRBM, N=%d)' % cfg['num_nodes'],
            'value': value, 'unit': 'it/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / (args.steps * args.repeats), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic', 'repeats': args.repeats, 'region_ms': [1e3 * v for v in region_s],
            'statistic': 'value = instances x (steps x repeats) / SUM of the timed regions (max over ranks each): %d CONSECUTIVE RTI steps after the 10 cold starts '
                         'and the warm-up, the transient of the first region included (BASELINE.md section 3.3)' % (args.steps * args.repeats),
            'solver_mode': MODE, 'solver_mode_is': MODE_TEXT[MODE],
            'median_region': {'value': n_inst * args.steps / main['median_region'], 'ms_per_step': 1e3 * main['median_region'] / args.steps,
                              'note': 'instances x steps / MEDIAN region: the settled rate of the same run (rounds 1-4 reported this as `value`)'},
            'collective': {'path': ('C-ABI: srbm_allgather_results (in-place ncclAllGather on the batch stream, communicator from srbm_rccl_comm_init_rank)'
                                    if coll['kind'] == 'cabi' else coll['kind'] if not DIST else 'torch.distributed all_gather_into_tensor (backend nccl = RCCL)'),
                           'note': coll['note']},
            'rccl_world_size': dist.get_world_size() if DIST else 1, 'per_rank_region_ms': per_rank_ms,
            'weak_scaling_efficiency': (value / (world * args.n1_baseline)) if args.n1_baseline > 0 else None,
            'config': {'workload': ('Config B: %d A1 SRBM MPC instances per GPU, N=20, dt=0.05, a1_configuration.yaml values, '
                                    '10 cold-start solves then open-loop RTI steps (state := node 1)' % B) if args.workload == 'B' else
                                   ('Config D: %d A1 SRBM MPC instances per GPU, N=50, dt=0.02, a1_config_distr_rejection.yaml values, push '
                                    'distribution on the initial momentum, 10 cold-start solves then open-loop RTI steps' % B) if args.workload == 'D' else
                                   ('Config E (SRBM stand-in for the dead centroidal MPC, no reference parity beyond the SRBM restatement): %d instances per GPU, '
                                    'N=40, dt=0.05, a1_configuration.yaml values, LARGE-capacity build (normal matrix in L2)' % B),
                       'batch_per_gpu': B, 'global_batch': n_inst, 'num_nodes': cfg['num_nodes'], 'parallelism': 'instances sharded x%d' % world,
                       'records_gathered': int(allrec.shape[0]), 'record_doubles': LD,
                       'all_solved': bool(q[2] == 0 and q[0] == 0), 'statuses_last_step': {int(k): int(v) for k, v in zip(*np.unique(status_all, return_counts=True))},
                       'timed_solves': int(q[1]), 'not_solved_in_timed_solves': int(q[2]), 'max_iter_in_timed_solves': int(q[3]),
                       'instances_with_a_solve_not_solved_rank0': [int(lo + b) for b in np.nonzero(acc_main[:, 2])[0][:64]],
                       'err_bits_all_timed_steps': int(q[0]) | int(np.bitwise_or.reduce(err_all)),
                       'mean_ipm_iterations': main['mean_iters'],
                       'solver': {'tol_gap': 1e-15, 'tol_feas': 1e-10, 'tol_step': mpc.solver_step_rule()[0], 'start_mu': mpc.solver_step_rule()[1],
                                  'solves_ended_by_step_rule': ctrs['step_rule'], 'lower_start_attempts': ctrs['low_tried'],
                                  'attempts_repeated_from_standard_start': ctrs['low_failed'], 'solves_counted': ctrs['solves'],
                                  'note': 'rank 0, all solves since the accumulators were cleared (timed regions; include/srbm_rti.h srbm_set_solver_step_rule)'}},
            'roofline': roof,
        }
        for mo, key in (('reference', 'reference_criterion'), ('lower_start', 'reference_criterion_lower_start'), ('step_rule', 'step_rule_mode'), ('one_region', 'one_region')):
            if mo in other:
                out[key] = other[mo]
        if d_stats is not None:
            out['config_d'] = d_stats
        if e_stats is not None:
            out['config_e'] = e_stats
        if steady is not None:
            out['steady_state'] = steady
        if gait_stats is not None:
            out['gait'] = gait_stats
        if cl_stats is not None:
            out['closed_loop'] = cl_stats
        if wbc_stats is not None:
            out['wbc'] = wbc_stats
        if world == 1 and not args.no_cpu_baseline and args.workload == 'B':
            try:
                out['cpu_baseline'] = cpu_baseline(cfg)
                cb = out['cpu_baseline']
                cb['criterion'] = "the reference's: Clarabel restatement to gap 1e-15 (clarabel_interface.cpp:165-175)"
                like = other['reference']['value'] if 'reference' in other else (value if MODE == 'reference' else None)
                cb['like_for_like'] = {'gpu_value_at_the_same_criterion_and_starting_point': like, 'gpu_value_headline_mode': value,
                                       'speedup_vs_one_thread': (like / cb['value']) if like and cb['value'] else None,
                                       'speedup_vs_all_cores': (like / cb['all_cores']['value']) if like and cb['all_cores']['value'] else None,
                                       'note': 'the CPU restatement runs Clarabel\'s starting point to Clarabel\'s criterion: the (0, 0) run is the like-for-like pair; the headline mode ends every solve by the same criterion from a lower starting point'}
            except Exception as e:            # the GPU line stands on its own
                out['cpu_baseline'] = {'value': None, 'unit': 'it/s', 'cores': 0, 'kind': 'port', 'sample': 'failed: %s' % e}
        print(json.dumps(out))
    if DIST:
        dist.barrier()
        if coll['comm'] is not None:
            mpc.rccl_comm_destroy(coll['comm'])
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
