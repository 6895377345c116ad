mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_c8.py tests/test_gpu_damped.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_entry_point.py -x -q > gpurun_out/r03/pytest_q.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03/pytest_q.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --steps 8 > gpurun_out/r03/bench_q.json 2> gpurun_out/r03/bench_q.err; echo "rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_q.json'))
print('fp32', l['value'], l['ms_per_step'])
b=l['bf16']; print('bf16', b['mode'], b['value'], b['ms_per_step'], b['roofline']['frac'])
PY
