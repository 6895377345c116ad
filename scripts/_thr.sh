cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/pmc_halo; mkdir -p $O
export IISEG_BF16_WINO_MIN_CIN=4096
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $O/a --output-format csv -- python3 $R/scripts/pmc_layer.py 128 128 211 bf16 64 119 > $O/a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O/b --output-format csv -- python3 $R/scripts/pmc_layer.py 128 128 211 bf16 64 119 > $O/b.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/c --output-format csv -- python3 $R/scripts/pmc_layer.py 128 128 211 bf16 64 119 > $O/c.log 2>&1
cd $O && python3 - <<'PY'
import csv, glob, collections
for d in 'abc':
    for f in glob.glob('%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_halo_bf16' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k, len(v), '%.4g' % (sum(v[1:]) / max(len(v) - 1, 1)))
PY
