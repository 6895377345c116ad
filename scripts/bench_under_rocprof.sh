set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/benchprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-recompute --no-two-streams --no-strict-f64 --no-configs --no-early-stop > $O/bench_under_rocprof.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
cd $R
python3 - <<'PY'
import json, glob, csv
d=json.loads(open('gpurun_out/r05/benchprof/bench_under_rocprof.json').read().strip().splitlines()[-1])
print('value', d['value'], 'avg launch ms (events):', d['roofline']['kernel'], d['roofline']['avg_launch_ms'])
for k in ('bf16','bf16x3'):
    if k in d:
        print(k, d[k]['value'], d[k]['roofline']['kernel'], d[k]['roofline']['avg_launch_ms'])
f=glob.glob('gpurun_out/r05/benchprof/run/**/*kernel_stats.csv',recursive=True)[0]
import shutil; shutil.copy(f,'gpurun_out/r05/benchprof/kernel_stats.csv')
for i,r in enumerate(csv.DictReader(open(f))):
    if i<8: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
