#!/bin/bash
# One SQ + GRBM counter pass (matrix-pipe busy, wave states) over any python command of the repo, summarised per
# kernel into a markdown table.  Usage: bash scripts/pmc_sq_cmd.sh <out.md> <title> <script.py> [args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/$1; title=$2; shift 2
O=$(mktemp -d /tmp/pmcsq.XXXXXX)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O --output-format csv -- python3 "$R/$1" "${@:2}" > $O/out.log 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
cd $R
python3 - "$O" "$out" "$title" <<'PY'
import csv, glob, sys, collections, re
d, out, title = sys.argv[1:4]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k); k = re.sub(r'^void ', '', k)
    per[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
rows = sorted(per.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE'])
L = ['# ' + title, '', 'One SQ + GRBM pass (scripts/pmc_sq_cmd.sh).  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); '
     'wave states as fractions of SQ_WAVE_CYCLES; share = of the GPU-active cycles of the command.', '',
     '| kernel | launches | share | matrix pipe busy | parked | issue stall | issuing | of which VALU |', '|---|---:|---:|---:|---:|---:|---:|---:|']
tot = sum(v['GRBM_GUI_ACTIVE'] for _, v in rows) or 1
for k, v in rows[:14]:
    gui = v['GRBM_GUI_ACTIVE'] / 8.0; wc = v['SQ_WAVE_CYCLES'] or 1
    L.append('| %s | %d | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |' % (k[:72], len(cnt[k]), v['GRBM_GUI_ACTIVE'] / tot,
             v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024) if gui else 0, v['SQ_WAIT_ANY'] / wc, v['SQ_WAIT_INST_ANY'] / wc,
             v['SQ_ACTIVE_INST_ANY'] / wc, v['SQ_ACTIVE_INST_VALU'] / wc))
open(out, 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
PY
rm -rf $O
