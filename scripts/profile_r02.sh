#!/bin/bash
# Round-2 profiling run (on the GPU box, from the repo root): rocprofv3 kernel traces of the bench
# workload in fp32 and with bf16 operands, the two HBM-traffic PMC passes, a HIP-API trace of the
# eager and the graph-replay loop.  Raw output under gpurun_out/r02/prof (scratch); summaries are
# made by scripts/make_profiles.py and committed under profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# two warm-up batches: the first primes the border stores, the second meets the steady-state windows
# (in bf16 mode: times the two 16-bit forms on them once); `make_profiles.py steady ... 6` drops both
B="$R/bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --no-bf16 --steps 3 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f32 --output-format csv -- python3 $B > $O/f32.json 2> $O/f32.err
IISEG_MMA=bf16 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bf16 --output-format csv -- python3 $B > $O/bf16.json 2> $O/bf16.err
B1="$R/bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --no-bf16 --no-roofline --steps 1 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $B1 > $O/fetch.json 2> $O/fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $B1 > $O/write.json 2> $O/write.err
IISEG_GRAPH=0 timeout -k 10 300 rocprofv3 --hip-trace -d $O/hip_eager --output-format csv -- python3 $B1 > $O/hip_eager.json 2> $O/hip_eager.err
IISEG_GRAPH=1 timeout -k 10 300 rocprofv3 --hip-trace -d $O/hip_graph --output-format csv -- python3 $B1 > $O/hip_graph.json 2> $O/hip_graph.err
# keep the merged-back output small: the per-dispatch traces are large
for d in fetch write; do python3 - "$O/$d" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
per = collections.OrderedDict()
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'], r['Counter_Name'])
    e = per.setdefault(k, [0, 0.0])
    if r['Dispatch_Id'] not in seen[k]:
        seen[k].add(r['Dispatch_Id']); e[0] += 1
    e[1] += float(r['Counter_Value'])
with open(sys.argv[1] + '/summary.csv', 'w') as o:
    o.write('Kernel_Name,Counter_Name,dispatches,Counter_Sum\n')
    for (k, c), (n, v) in per.items():
        o.write('"%s",%s,%d,%.6g\n' % (k.replace('"', "'"), c, n, v))
PY
rm -f $O/$d/*/*counter_collection.csv; done
for d in hip_eager hip_graph; do python3 - "$O/$d" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*hip_api_trace.csv', recursive=True)[0]
c = collections.Counter(r['Function'] for r in csv.DictReader(open(f)))
open(sys.argv[1] + '/api_counts.txt', 'w').write('\n'.join('%s %d' % kv for kv in c.most_common()))
PY
rm -f $O/$d/*/*hip_api_trace.csv; done
echo profiling done
