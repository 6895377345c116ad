# per-kernel time table of a python command under rocprofv3 --kernel-trace --stats
# Usage: bash scripts/kstats.sh <out.csv> <script.py> [args...]   (run from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/$1; shift
d=$(mktemp -d /tmp/kstats.XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 "$R/$1" "${@:2}" > $d/out.log 2>&1 || { tail -5 $d/out.log; exit 1; }
tail -2 $d/out.log
cd $R
python3 - $d $out <<'PY'
import csv, glob, sys, re
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(out, 'w') as o:
    o.write('kernel,calls,total_ms,avg_ms,percent\n')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'\(.*', '', name); name = re.sub(r'^void ', '', name)
        o.write('%s,%s,%.3f,%.4f,%.2f\n' % (name[:90].replace(',', ';'), r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                           float(r['AverageNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
    o.write('TOTAL,,%.3f,,100\n' % (tot / 1e6))
PY
rm -rf $d
head -22 $out
