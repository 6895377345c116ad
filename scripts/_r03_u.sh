mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_c8.py -x -q -s -k random > gpurun_out/r03/c8_fuzz.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r03/c8_fuzz.log
IISEG_C8_TALL=1 timeout -k 10 600 python -m pytest tests/test_gpu_c8.py -x -q -s -k random > gpurun_out/r03/c8_fuzz_tall.log 2>&1; echo "tall rc=$?"; tail -3 gpurun_out/r03/c8_fuzz_tall.log
