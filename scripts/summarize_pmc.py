#!/usr/bin/env python3
"""profiles/<round>/ from the raw rocprofv3 output of scripts/collect_profiles.sh (gpurun_out/prof/):
   python scripts/summarize_pmc.py r02
Writes bench_kernel_stats.csv (the --stats kernel summary), timed_region.txt (durations of the timed-region launches of the
dominant kernel from the kernel trace, beside the live HIP-event figure of the same run), bench_*.json (the bench lines of the
runs) and pmc_summary.json: every PMC pass per launch of the timed region -- HBM traffic with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md, matrix-pipe busy fraction, LDS bank conflicts, wave occupancy -- stamped with the
sha of the kernel sources it was measured on (bench.py quotes it only for those sources)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rnd = sys.argv[1] if len(sys.argv) > 1 else bench.PROFILE_ROUND
src = os.path.join(ROOT, 'gpurun_out', 'prof')
dst = os.path.join(ROOT, 'profiles', rnd)
os.makedirs(dst, exist_ok=True)
KERNEL = 'srbm_rti_fused'


def find(sub, pat):
    hits = sorted(glob.glob(os.path.join(src, sub, '**', pat), recursive=True))
    return hits[0] if hits else None


def bench_line(name):
    try:
        return json.loads(open(os.path.join(src, name)).read().strip().splitlines()[-1])
    except Exception:
        return None


shutil.copy(find('trace', '*kernel_stats.csv'), os.path.join(dst, 'bench_kernel_stats.csv'))
f3 = find('trace_f3', '*kernel_stats.csv')
if f3:
    # control-tick kernels (row f3): average duration per call from the kernel trace, next to the bench's own per-tick figures
    want = ('srbm_k_targets_from_traj', 'srbm_k_qp_control', 'srbm_k_eval_trajectory')
    rows3 = [r for r in csv.DictReader(open(f3)) if any(w in r['Name'] for w in want)]
    b3 = bench_line('bench_f3_under_rocprof.json')
    with open(os.path.join(dst, 'f3_kernel_stats.txt'), 'w') as fh:
        fh.write('control-tick kernels of one batch of 256 instances, rocprofv3 --kernel-trace --stats of bench.py (fourth segment):\n')
        for r in rows3:
            fh.write('%-28s calls %3s  avg %.3f ms  min %.3f  max %.3f\n' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6, float(r['MaxNs']) / 1e6))
        if b3 and 'wbc' in b3:
            w = b3['wbc']
            fh.write('bench.py of the same run: %.3f ms per tick (targets %.3f, whole-body QP %.3f through the host-pointer entries), targets not ok %d, QPs not solved %d\n' %
                     (w['ms_per_tick_of_the_batch'], w['ms_per_tick_targets_only'], w.get('ms_per_tick_qp_control_only', float('nan')), w['targets_not_ok'], w['qp_not_solved']))
            if 'device_resident' in w:
                fh.write('   ... and the same ticks device-resident (device-pointer entries on the batch\'s stream, no copy / synchronisation inside a tick; the kernel calls above '
                         'include them): %.3f ms per tick, targets not ok %d, QPs not solved %d\n' % (w['device_resident']['ms_per_tick_of_the_batch'],
                         w['device_resident']['targets_not_ok'], w['device_resident']['qp_not_solved']))
for name in ('bench_unprofiled.json', 'bench_under_rocprof.json'):
    shutil.copy(os.path.join(src, name), os.path.join(dst, name))
ub, pb = bench_line('bench_unprofiled.json'), bench_line('bench_under_rocprof.json')
steps, repeats = pb['steps'], pb['repeats']
rows = [r for r in csv.DictReader(open(find('trace', '*kernel_trace.csv'))) if KERNEL in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# launches of the fused kernel in the profiled command, in order: [0] the warm-up launch, [1 .. repeats] the TIMED regions of the headline, then (steady_state
# of the bench line) one untimed advance and `repeats` more regions from step 125 on.  (Rounds 3-4 took the LAST `repeats` launches here -- the steady-state
# ones -- and called them the timed regions: their timed_region.txt / pmc_summary.json describe the settled protocol, not the headline's launches.)
timed = rows[1:1 + repeats]
ms = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in timed]
ms_steady = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows[2 + repeats:2 + 2 * repeats]]
with open(os.path.join(dst, 'timed_region.txt'), 'w') as f:
    f.write('%s: %d launches in the trace (1 warm-up launch of %d steps + %d timed regions of %d steps + 1 untimed advance + %d steady-state regions)\n' % (KERNEL, len(rows), pb['warmup'], repeats, steps, len(ms_steady)))
    med = sorted(ms)[len(ms) // 2]
    f.write('timed-region launches, kernel trace:  %s ms   median %.3f ms = %.4f ms per RTI step (mean %.3f; the first region follows the cold start)\n' % (' '.join('%.3f' % v for v in ms), med, med / steps, sum(ms) / len(ms)))
    f.write('live HIP-event figures of the same run (bench.py roofline.launch_ms_all): %s ms; their MEAN is what `roofline` describes (value = sum of the regions): %.3f ms\n' % (' '.join('%.3f' % v for v in pb['roofline'].get('launch_ms_all', [])), pb['roofline']['avg_launch_ms']))
    if ms_steady:
        f.write('steady-state regions (from step %d on), kernel trace: %s ms   mean %.3f ms = %.4f ms per RTI step\n' % (pb.get('steady_state', {}).get('first_step', 125), ' '.join('%.3f' % v for v in ms_steady), sum(ms_steady) / len(ms_steady), sum(ms_steady) / len(ms_steady) / steps))
    f.write('unprofiled run: avg_launch_ms %.3f, ms_per_step %.4f, value %.0f it/s\n' % (ub['roofline']['avg_launch_ms'], ub['ms_per_step'], ub['value']))


MED_IDX = sorted(range(len(ms)), key=lambda i: ms[i])[len(ms) // 2]


def counters(sub):
    """per counter: mean over the timed-region dispatches of the kernel (sum over the agents / XCDs / SEs as rocprofv3 reports them)"""
    path = find(sub, '*counter_collection.csv')
    if not path:
        return {}
    per = {}
    for r in csv.DictReader(open(path)):
        if KERNEL in r['Kernel_Name']:
            d = per.setdefault(r['Counter_Name'], {})
            d[int(r['Dispatch_Id'])] = d.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
    # what `value` and `roofline` describe since round 5: ALL timed launches (mean per launch; same deterministic workload in every pass)
    out = {}
    for c, d in per.items():
        ids = sorted(d)[1:1 + repeats]           # (dispatch [0] is the warm-up launch; the steady-state regions come after the timed ones)
        out[c] = sum(d[i] for i in ids) / len(ids)
    return out


F, W, S1, S2 = counters('pmc_fetch'), counters('pmc_write'), counters('pmc_sq1'), counters('pmc_sq2')
summ = {'kernel': KERNEL, 'round': rnd, 'kernel_source_sha': bench.kernel_source_sha(), 'steps_per_launch': steps, 'timed_launches': repeats,
        'command': 'rocprofv3 --pmc <one pass per counter group, no other tracing> -- python3 bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0 --wbc-ticks 0',
        'launch_described': 'mean over the %d timed launches (one per region): counters, duration and the roofline object of the bench line all refer to it' % len(ms),
        'kernel_trace_ms_per_launch': sum(ms) / len(ms), 'kernel_trace_ms_each': ms, 'hip_event_ms_per_launch_same_run': pb['roofline']['avg_launch_ms'],
        'raw_counters_per_launch': {**F, **W, **S1, **S2}}
if 'FETCH_SIZE' in F and 'WRITE_SIZE' in W:
    hbm = (2.0 * F['FETCH_SIZE'] + W['WRITE_SIZE']) * 1024.0
    summ.update({'FETCH_SIZE_KB_per_launch': F['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': W['WRITE_SIZE'],
                 'hbm_correction': 'gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section; calibrated there for 16 B/lane '
                                   'streaming reads, this kernel reads 8 B/lane: an upper bound); WRITE_SIZE as reported; both sit on the memory side of L2 and '
                                   'include Infinity-Cache hits',
                 'hbm_bytes_per_launch': hbm, 'hbm_bytes_per_step': hbm / steps,
                 'algorithmic_bytes_per_step': 256 * 12.8e3, 'traffic_over_algorithmic': hbm / steps / (256 * 12.8e3)})
if 'SQ_WAVE_CYCLES' in S1:
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count
    # cycles (guide, 's_memtime tick vs SQ PMC units'); GRBM_GUI_ACTIVE is summed over the 8 XCDs
    gui = S1.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    summ.update({'gpu_cycles_per_launch': gui,
                 'effective_clock_GHz': gui / (sum(ms) / len(ms) * 1e6) if gui else None,
                 'mfma_busy_cycles_per_launch': S1.get('SQ_VALU_MFMA_BUSY_CYCLES'),
                 # 256 CUs x 4 SIMDs each could be busy for every GPU cycle of the launch
                 'mfma_busy_frac': S1.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 256 * 4) if gui else None,
                 'occupancy_waves_per_cu': 4.0 * S1['SQ_WAVE_CYCLES'] / (gui * 256) if gui else None,
                 'wave_cycles_split': {k: S1[k] / S1['SQ_WAVE_CYCLES'] for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if k in S1}})
if 'SQ_LDS_BANK_CONFLICT' in S2:
    summ.update({'lds_bank_conflict_frac': S2['SQ_LDS_BANK_CONFLICT'] / S2['SQ_LDS_IDX_ACTIVE'] if S2.get('SQ_LDS_IDX_ACTIVE') else None,
                 'lds_active_frac_of_wave_cycles': (S2.get('SQ_ACTIVE_INST_LDS', 0.0) / S1['SQ_WAVE_CYCLES']) if S1.get('SQ_WAVE_CYCLES') else None,
                 'valu_busy_frac': (S2.get('SQ_ACTIVE_INST_VALU', 0.0) / S1['SQ_WAVE_CYCLES']) if S1.get('SQ_WAVE_CYCLES') else None})
# ---- round 3: the other workloads / segments ----
def kernel_stats(sub, out_name, title, bench_json=None):
    f = find(sub, '*kernel_stats.csv')
    if not f:
        return
    rows_ = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, out_name), 'w') as fh:
        fh.write('# %s\n' % title)
        b_ = bench_line(bench_json) if bench_json else None
        if b_:
            fh.write('# bench line of the same run: value %.0f %s, ms_per_step %.4f, workload: %s\n' % (b_['value'], b_['unit'], b_['ms_per_step'], b_['config']['workload'][:120]))
            if 'gait' in b_:
                fh.write('# gait segment of the same run: %s\n' % json.dumps({k: b_['gait'][k] for k in ('ms_per_step', 'rti_solves_per_s_incl_line_search', 'gait_steps_per_s', 'err_bits_all_steps', 'not_solved_all_steps')}))
        fh.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
        for r in rows_:
            fh.write('%s,%s,%s,%s,%s,%s,%s\n' % (r['Name'].split('(')[0], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']))


kernel_stats('trace_D', 'D_kernel_stats.csv', 'rocprofv3 --kernel-trace --stats -- python3 bench.py --workload D --no-cpu-baseline --closed-loop-steps 0 (512 instances, N = 50: standard kernel set, two rounds of 256 workgroups)', 'bench_D_under_rocprof.json')
kernel_stats('trace_E', 'E_kernel_stats.csv', 'rocprofv3 --kernel-trace --stats -- python3 bench.py --workload E --no-cpu-baseline --closed-loop-steps 0 (128 instances, N = 40: LARGE build)', 'bench_E_under_rocprof.json')
kernel_stats('trace_gait', 'gait_kernel_stats.csv', 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --closed-loop-steps 0 --wbc-ticks 0 (Config B region + the gait segment: 30 steps, 6 gait steps, 6 line searches)', 'bench_gait_under_rocprof.json')


def counters_of(sub, kernel_sub):
    path = find(sub, '*counter_collection.csv')
    if not path:
        return {}
    per, n = {}, {}
    for r in csv.DictReader(open(path)):
        if kernel_sub in r['Kernel_Name']:
            per[r['Counter_Name']] = per.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            n.setdefault(r['Counter_Name'], set()).add(int(r['Dispatch_Id']))
    return {c: v / max(1, len(n[c])) for c, v in per.items()}, {c: len(v) for c, v in n.items()}


extra = {}
cd = counters_of('pmc_D_sq1', 'srbm_rti_queued')          # (batches larger than the chip run on the step queues since round 5: srbm_fused.hiph)
if cd:
    c, nd = cd
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    extra['config_D'] = {'kernel': 'srbm_rti_queued_long (512 instances: a resident grid of 256 workgroups takes (instance, step) items from the per-XCD queues; the cold starts and the warm-up of the run are not in it)', 'dispatches_averaged': nd, 'raw_counters_per_launch': c,
        'waves_launched_per_launch': c.get('SQ_WAVES'),
        'occupancy_waves_per_cu': 4.0 * c['SQ_WAVE_CYCLES'] / (gui * 256) if gui and 'SQ_WAVE_CYCLES' in c else None,
        'mfma_busy_frac': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 256 * 4) if gui else None,
        'wave_cycles_split': {k: c[k] / c['SQ_WAVE_CYCLES'] for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if k in c and c.get('SQ_WAVE_CYCLES')}}
for kn in ('srbm_k_targets_from_traj', 'srbm_k_qp_control'):
    c1 = counters_of('pmc_f3_sq1', kn); c2 = counters_of('pmc_f3_sq2', kn)
    if c1 and c2:
        a, na = c1; b_, nb = c2
        gui = a.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        extra[kn] = {'dispatches_averaged': na, 'raw_counters_per_launch': {**a, **b_},
                     'occupancy_waves_per_cu': 4.0 * a['SQ_WAVE_CYCLES'] / (gui * 256) if gui and 'SQ_WAVE_CYCLES' in a else None,
                     'wave_cycles_split': {k: a[k] / a['SQ_WAVE_CYCLES'] for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if k in a and a.get('SQ_WAVE_CYCLES')},
                     'valu_instructions_per_wave': b_.get('SQ_INSTS_VALU', 0.0) / a['SQ_WAVES'] if a.get('SQ_WAVES') else None,
                     'lds_bank_conflict_frac': b_['SQ_LDS_BANK_CONFLICT'] / b_['SQ_LDS_IDX_ACTIVE'] if b_.get('SQ_LDS_IDX_ACTIVE') else None}
if os.path.exists(os.path.join(src, 'wbc_phase_shares.txt')):
    shutil.copy(os.path.join(src, 'wbc_phase_shares.txt'), os.path.join(dst, 'wbc_phase_shares.txt'))
if os.path.exists(os.path.join(src, 'phase_shares_reference_criterion.txt')):
    shutil.copy(os.path.join(src, 'phase_shares_reference_criterion.txt'), os.path.join(dst, 'ipm_phase_shares_reference_criterion.txt'))
if os.path.exists(os.path.join(src, 'phase_shares.txt')):
    shutil.copy(os.path.join(src, 'phase_shares.txt'), os.path.join(dst, 'ipm_phase_shares.txt'))
    # the two numbers rounds are compared by: cycles of one IPM iteration and the share of the dense phase, from the coarse stamps
    # of the diagnostic build (scripts/dev_prof.py: last solve of instance 0 after 4 RTI steps)
    try:
        txt = open(os.path.join(src, 'phase_shares.txt')).read().splitlines()
        it_line = [l for l in txt if l.startswith('iters ')][0].split()
        iters, ticks = float(it_line[1]), float(it_line[-1])
        coarse = txt[:next((i for i, l in enumerate(txt) if l.startswith('fine stamps')), len(txt))]
        share = {l[:18].strip(): float(l.split()[-1].rstrip('%')) for l in coarse if l.rstrip().endswith('%') and len(l.split()) >= 3}
        summ['ipm_iteration'] = {'source': 'profiles/%s/ipm_phase_shares.txt (diagnostic build with stamps: shares, not times)' % rnd, 'solver_settings': txt[0],
                                 'iterations_of_the_solve': iters, 'stamp_ticks_per_iteration': ticks / max(iters, 1.0),
                                 'cholesky_incl_rank2_and_inverse_share_pct': share.get('Cholesky'), 'coarse_shares_pct': share}
    except Exception as e:
        summ['ipm_iteration'] = {'error': str(e)}
if extra:
    summ['other_kernels'] = extra
json.dump(summ, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1)
print(open(os.path.join(dst, 'timed_region.txt')).read())
print(json.dumps({k: v for k, v in summ.items() if k != 'raw_counters_per_launch'}, indent=1))
