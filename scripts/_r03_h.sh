mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
IISEG_C8_TALL=1 timeout -k 10 600 python -m pytest tests/test_gpu_c8.py -x -q > gpurun_out/r03/c8_tall.log 2>&1; echo "tall rc=$?"; tail -3 gpurun_out/r03/c8_tall.log
timeout -k 10 600 python -m pytest tests/test_gpu_c8.py -x -q > gpurun_out/r03/c8_def.log 2>&1; echo "default rc=$?"; tail -3 gpurun_out/r03/c8_def.log
IISEG_C8_TALL=0 timeout -k 10 300 python scripts/bench_c8.py > gpurun_out/r03/bench_c8_t0.log 2>&1
timeout -k 10 300 python scripts/bench_c8.py > gpurun_out/r03/bench_c8_t1.log 2>&1
paste <(awk '{print $1,$2,$3,$4,$11,$12}' gpurun_out/r03/bench_c8_t0.log) <(awk '{print $11,$12,$13,$14,$15}' gpurun_out/r03/bench_c8_t1.log)
