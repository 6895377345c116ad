mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_o.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r03/pytest_o.log; grep -h "steps: float64 vs" gpurun_out/r03/pytest_o.log | head -20
