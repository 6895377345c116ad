#!/bin/bash
# Round-5 profiling run (same recipe as rounds 3 and 4) (on the GPU box, from the repo root).  rocprofv3 kernel traces of one leg at
# a time (scripts/run_leg.py: 2 warm-up batches + 3 batches) and PMC passes -- every --pmc pass on
# its own, with --kernel-trace only, and within the per-block counter slots of gfx950
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8, TCC 4 with FETCH_SIZE = 3 and WRITE_SIZE = 2,
# GRBM 2).  Raw output under gpurun_out/r05/prof (scratch); summaries by scripts/make_profiles.py.
# Usage: profile_r05.sh [legs...]   legs: f32 bf16c8 bf16x3 f64 pmc_f32 pmc_bf16c8 pmcsq_<leg>[:batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEGS=${@:-f32 bf16c8 f64}
summ() {  # per-kernel sums of a counter_collection.csv -> summary.csv, raw file removed
python3 - "$1" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: sys.exit(0)
per = collections.OrderedDict(); seen = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
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
rm -f $1/*/*counter_collection.csv
}
for leg in $LEGS; do
  case $leg in
    f32|bf16c8|bf16|bf16x3)
      timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$leg --output-format csv -- python3 $R/scripts/run_leg.py $leg 64 3 > $O/$leg.out 2> $O/$leg.err || exit 1 ;;
    f64)
      timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f64 --output-format csv -- python3 $R/scripts/run_leg.py f64 16 2 > $O/f64.out 2> $O/f64.err || exit 1 ;;
    pmcsq_*)   # SQ-side counters only (one pass), e.g. pmcsq_f64:16
      m=${leg#pmcsq_}; b=64; case $m in *:*) b=${m#*:}; m=${m%%:*};; esac
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/${m}_sq --output-format csv -- python3 $R/scripts/run_leg.py $m $b 1 > $O/${m}_sq.out 2> $O/${m}_sq.err || exit 1
      summ $O/${m}_sq ;;
    pmc_*)
      m=${leg#pmc_}
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${m}_fetch --output-format csv -- python3 $R/scripts/run_leg.py $m 64 1 > $O/${m}_fetch.out 2> $O/${m}_fetch.err || exit 1
      summ $O/${m}_fetch
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${m}_write --output-format csv -- python3 $R/scripts/run_leg.py $m 64 1 > $O/${m}_write.out 2> $O/${m}_write.err || exit 1
      summ $O/${m}_write
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/${m}_sq --output-format csv -- python3 $R/scripts/run_leg.py $m 64 1 > $O/${m}_sq.out 2> $O/${m}_sq.err || exit 1
      summ $O/${m}_sq ;;
  esac
done
echo profiling done
