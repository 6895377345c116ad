mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_f64.py tests/test_gpu_damped.py -x -q -s > gpurun_out/r03/f64_1.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r03/f64_1.log
timeout -k 10 300 python scripts/bench_configs.py c2_f64 2 2>/dev/null | tail -1
IISEG_WINO_F64=0 timeout -k 10 300 python scripts/bench_configs.py c2_f64 2 2>/dev/null | tail -1
