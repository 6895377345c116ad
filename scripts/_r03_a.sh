set -x
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_damped.py -x -q -s > gpurun_out/r03/damped1.log 2>&1; echo "damped rc=$?" >> gpurun_out/r03/damped1.log
timeout -k 10 300 python scripts/sensitivity.py c2d > gpurun_out/r03/sens_c2d.log 2>&1
tail -5 gpurun_out/r03/damped1.log
