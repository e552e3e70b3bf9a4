#!/usr/bin/env python3
"""profiles/<round>/ from the raw rocprofv3 output of scripts/collect_profiles.sh (gpurun_out/prof/):
   python scripts/summarize_pmc.py r01
Writes bench_kernel_stats.csv (the --stats kernel summary), k3_timed_region.txt (duration of the timed-region launch of the
dominant kernel from the kernel trace), bench_*.json (the bench lines of the runs) and k3_pmc_traffic.json (the two PMC
passes, per launch of the timed region, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', 'prof')
dst = os.path.join(ROOT, 'profiles', rnd)
os.makedirs(dst, exist_ok=True)
KERNEL = 'srbm_rti_fused'


def find(sub, pat):
    hits = sorted(glob.glob(os.path.join(src, sub, '**', pat), recursive=True))
    if not hits:
        raise SystemExit('missing %s/%s' % (sub, pat))
    return hits[0]


shutil.copy(find('trace', '*kernel_stats.csv'), os.path.join(dst, 'bench_kernel_stats.csv'))
for name in ('bench_unprofiled.json', 'bench_under_rocprof.json'):
    shutil.copy(os.path.join(src, name), os.path.join(dst, name))
rows = [r for r in csv.DictReader(open(find('trace', '*kernel_trace.csv'))) if KERNEL in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-1]
ms = (int(last['End_Timestamp']) - int(last['Start_Timestamp'])) / 1e6
bench = json.loads(open(os.path.join(src, 'bench_under_rocprof.json')).read().strip().splitlines()[-1])
steps = bench['steps']
open(os.path.join(dst, 'k3_timed_region.txt'), 'w').write(
    '%s: %d launches (warm-up, timed region of %d steps); timed-region launch: %.3f ms = %.3f ms per RTI step; '
    'live HIP-event figure of the same run: %.3f ms\n' % (KERNEL, len(rows), steps, ms, ms / steps, bench['roofline']['avg_launch_ms']))


def counter(sub, cname):
    tot = {}
    for r in csv.DictReader(open(find(sub, '*counter_collection.csv'))):
        if KERNEL in r['Kernel_Name'] and r['Counter_Name'] == cname:
            tot.setdefault(r['Dispatch_Id'], 0.0)
            tot[r['Dispatch_Id']] += float(r['Counter_Value'])
    ids = sorted(tot, key=int)
    return tot[ids[-1]], len(ids)          # the last dispatch is the timed region


fetch_kb, nl = counter('pmc_fetch', 'FETCH_SIZE')
write_kb, _ = counter('pmc_write', 'WRITE_SIZE')
hbm = (2.0 * fetch_kb + write_kb) * 1024.0
json.dump({
    'kernel': KERNEL, 'steps_per_launch': steps, 'launches_seen': nl,
    'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes, no other tracing) -- python3 bench.py --no-cpu-baseline --gait-steps 0 --closed-loop-steps 0',
    'FETCH_SIZE_KB_per_launch': fetch_kb, 'WRITE_SIZE_KB_per_launch': write_kb,
    'correction': 'gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section; calibrated there for '
                  '16 B/lane streaming reads, this kernel reads 8 B/lane: upper bound); WRITE_SIZE as reported',
    'hbm_bytes_per_launch': hbm, 'hbm_bytes_per_rti_step': hbm / steps,
    'note': 'counters sit on the memory side of L2 and include Infinity-Cache hits',
}, open(os.path.join(dst, 'k3_pmc_traffic.json'), 'w'), indent=1)
print(open(os.path.join(dst, 'k3_timed_region.txt')).read())
print('FETCH %.1f MB/step  WRITE %.1f MB/step  corrected HBM %.1f MB/step' % (fetch_kb / 1024 / steps, write_kb / 1024 / steps, hbm / 1e6 / steps))
