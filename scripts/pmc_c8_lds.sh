# rocprofv3 PMC of single C8 layers: the LDS side (bank conflicts, busy cycles) in one pass.
# Usage: bash scripts/pmc_c8_lds.sh <outdir> layer...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_BUSY_CU_CYCLES -d $O/l_$L --output-format csv -- python3 $R/scripts/c8_layer.py $L 3 > $O/l_$L.log 2>&1 || { tail -5 $O/l_$L.log; exit 1; }
done
cd $R
python3 - $O "$@" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
for L in sys.argv[2:]:
    tot = collections.defaultdict(float)
    for f in glob.glob('%s/l_%s/**/*counter_collection.csv' % (O, L), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_c8_' not in r['Kernel_Name']: continue
            tot[r['Counter_Name']] += float(r['Counter_Value'])
    w = tot['SQ_WAVES']
    print(L, {k: round(v / w, 1) for k, v in tot.items() if k != 'SQ_WAVES'}, 'waves', w)
PY
for L in "$@"; do rm -rf $O/l_$L; done
