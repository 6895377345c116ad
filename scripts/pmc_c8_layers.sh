# rocprofv3 PMC of single C8 layers: cycles / waits, then the instruction mix (separate passes).
# Usage: bash scripts/pmc_c8_layers.sh <outdir> layer...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $O/a_$L --output-format csv -- python3 $R/scripts/c8_layer.py $L 3 ${C8_LAYER_BATCH:-64} > $O/a_$L.log 2>&1 || { tail -5 $O/a_$L.log; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS -d $O/b_$L --output-format csv -- python3 $R/scripts/c8_layer.py $L 3 ${C8_LAYER_BATCH:-64} > $O/b_$L.log 2>&1 || { tail -5 $O/b_$L.log; exit 1; }
done
cd $R
python3 - $O "$@" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
for L in sys.argv[2:]:
    tot = collections.defaultdict(float); n = 0
    for part in 'ab':
        for f in glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, part, L), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv_c8_' not in r['Kernel_Name']: continue
                tot[part + r['Counter_Name']] += float(r['Counter_Value'])
    wa, wb = tot['aSQ_WAVES'], tot['bSQ_WAVES']
    print(L, 'waves/launch %.0f' % (wa / 5))
    # SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles; BUSY_CYCLES per SE..; report per wave
    v = [4 * tot['a' + c] / wa for c in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU')]
    print('   per wave (cycles): wave %.0f  wait_any %.0f  wait_inst %.0f  active_any %.0f  active_valu %.0f  mfma_busy %.0f'
          % (tuple(v) + (tot['aSQ_VALU_MFMA_BUSY_CYCLES'] / wa,)))
    print('   per wave insts: VALU %.0f SALU %.0f MFMA %.0f VMEM %.0f LDS %.0f BRANCH %.0f   active_lds %.0f'
          % tuple(tot['b' + c] / wb for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_VMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_BRANCH', 'SQ_ACTIVE_INST_LDS')))
PY
for L in "$@"; do rm -rf $O/a_$L $O/b_$L; done
