set -x
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03/pytest_b.log 2>&1; echo "rc=$?" >> gpurun_out/r03/pytest_b.log
tail -25 gpurun_out/r03/pytest_b.log
