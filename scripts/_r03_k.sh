mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r03/pytest_k.log
for c in "fcn8dae 2" "fcn8dae 2 bf16" "c3 2" "c3 2 bf16" "c3 2 bf16c8" "c4 2" "c4 2 bf16c8" "c5 2" "c5 2 bf16c8"; do timeout -k 10 200 python scripts/bench_configs.py $c 2>/dev/null | tail -1; done > gpurun_out/r03/bench_configs.log; cat gpurun_out/r03/bench_configs.log
