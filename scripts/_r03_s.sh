mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
IISEG_C8_TALL=1 timeout -k 10 600 python -m pytest tests/test_gpu_c8.py -x -q > gpurun_out/r03/c8_tall.log 2>&1; echo "tall rc=$?"; tail -2 gpurun_out/r03/c8_tall.log
timeout -k 10 600 python -m pytest tests/test_gpu_c8.py tests/test_gpu_damped.py -x -q > gpurun_out/r03/c8_def.log 2>&1; echo "default rc=$?"; tail -2 gpurun_out/r03/c8_def.log
timeout -k 10 300 python scripts/bench_c8.py > gpurun_out/r03/bench_c8_e.log 2>&1
awk '{print $1,$2,$3,$4,$11,$12,$13,$14,$15}' gpurun_out/r03/bench_c8_e.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --steps 8 > gpurun_out/r03/bench_s.json 2> gpurun_out/r03/bench_s.err; echo "rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_s.json'))
b=l['bf16']; print('fp32', l['value'], 'bf16', b['mode'], b['value'], b['ms_per_step'], b['roofline']['frac'])
PY
