mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --no-bf16 --steps 8 --streams 2,4,8 > gpurun_out/r03/bench_w.json 2> gpurun_out/r03/bench_w.err; echo "rc=$?"; tail -3 gpurun_out/r03/bench_w.err
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_w.json'))
print('fp32', l['value'], l['ms_per_step'], 'two_streams', l.get("concurrent_streams"))
PY
