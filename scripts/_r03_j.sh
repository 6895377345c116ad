mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
bash scripts/profile_r03.sh pmcsq_f64:16 || { tail -5 gpurun_out/r03/prof/*.err; exit 1; }
grep "conv_taps_f64" gpurun_out/r03/prof/f64_sq/summary.csv | sed 's/.*conv_taps_f64_kernel<\([^>]*\)>[^,]*,/\1 /' | head -40
