cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/p_x -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dev_k2time.py > /dev/null 2>&1; python3 -c "
import csv,glob
f=glob.glob('/tmp/p_x/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('k2_condense','k4_update','k1_assemble')): print(n[:20], r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3))
"
