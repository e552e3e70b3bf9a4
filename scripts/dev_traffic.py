"""Where the HBM-side traffic of an RTI step comes from: the same protocol as the bench (Config B, 10 cold starts, 5 warm-up steps) with the TIMED steps
launched kernel by kernel (srbm_rti_advance_unfused), to be run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`:
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/traffic/fetch -o pmc --output-format csv -- python3 scripts/dev_traffic.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/traffic/write -o pmc --output-format csv -- python3 scripts/dev_traffic.py
    python3 scripts/dev_traffic.py --summarize gpurun_out/traffic
The summary adds the counter per kernel name over the last `steps` launches of each kernel (KB as rocprofv3 reports them; FETCH_SIZE x 2 on gfx950)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 20
if len(sys.argv) > 2 and sys.argv[1] == '--summarize':
    for name, mult in (('fetch', 2.0), ('write', 1.0)):
        files = glob.glob(os.path.join(sys.argv[2], name, '**', '*counter_collection.csv'), recursive=True)
        if not files: print(name, ': no counter file'); continue
        per = {}
        for row in csv.DictReader(open(files[0])):
            per.setdefault(row['Kernel_Name'].split('(')[0], []).append(float(row['Counter_Value']))
        print('%s (KB per instance-step, x %.0f correction applied; last %d launches of each kernel, 256 instances)' % (name, mult, STEPS))
        tot = 0
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-STEPS:])):
            if not k.startswith('srbm_k') or len(v) < STEPS: continue        # (the per-phase kernels of the timed steps only: the warm-up's fused launch, fills and copies left out)
            s = sum(v[-STEPS:]) * mult / (STEPS * 256); tot += s
            print('  %-40s launches %4d  %10.1f' % (k[:40], len(v), s))
        print('  %-40s %26.1f' % ('sum', tot))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
from srbm_loader import workloads
cfg = host.load_config()
B = 256
states, ees = zip(*[workloads.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
g.set_solver_step_rule(0.0, 0.1)
for _ in range(10): g.create_initial_run(states, ees)
g.rti_advance(0, 5); g.synchronize()
g.rti_advance_unfused(5, STEPS); g.synchronize()
print('statuses', dict(zip(*np.unique(g.status()[0], return_counts=True))), 'iters', g.stats()[:, 4].mean())
