set -x
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 600 python scripts/bench_c8.py > gpurun_out/r03/bench_c8_1.log 2>&1; echo "rc=$?" >> gpurun_out/r03/bench_c8_1.log
tail -30 gpurun_out/r03/bench_c8_1.log
