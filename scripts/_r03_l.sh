mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
bash scripts/profile_r03.sh bf16c8 f32 pmc_bf16c8 pmc_f32 || { tail -5 gpurun_out/r03/prof/*.err; exit 1; }
for l in f32 bf16c8; do python scripts/make_profiles.py steady gpurun_out/r03/prof/$l gpurun_out/r03/stats_$l.csv; rm -rf gpurun_out/r03/prof/$l; done
rm -rf gpurun_out/r03/prof/*/runc gpurun_out/r03/prof/*/*/*kernel_trace.csv
ls gpurun_out/r03/prof/*
head -8 gpurun_out/r03/stats_bf16c8.csv
