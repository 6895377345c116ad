set -x
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_c8.py tests/test_gpu_damped.py -x -q -s > gpurun_out/r03/c8_2.log 2>&1; echo "rc=$?" >> gpurun_out/r03/c8_2.log
tail -40 gpurun_out/r03/c8_2.log
