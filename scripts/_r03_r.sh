mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
bash scripts/profile_r03.sh bf16c8 || { tail -5 gpurun_out/r03/prof/*.err; exit 1; }
python scripts/make_profiles.py steady gpurun_out/r03/prof/bf16c8 gpurun_out/r03/stats_bf16c8_b.csv > /dev/null; rm -rf gpurun_out/r03/prof/bf16c8
cat gpurun_out/r03/stats_bf16c8_b.csv
