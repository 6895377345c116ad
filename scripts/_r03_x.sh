mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_x.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03/pytest_x.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03/smoke.log
/usr/bin/time -v timeout -k 10 1000 python bench.py > gpurun_out/r03/bench_full_2.json 2> gpurun_out/r03/bench_full_2.err; echo "bench rc=$?"; grep "Elapsed (wall" gpurun_out/r03/bench_full_2.err
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_full_2.json'))
print('fp32', l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['whole_path']['frac'], l['concurrent_streams'])
for k in ('per_batch_only','full_recompute','strict_f64'): print(k, l[k]['value'], l[k]['ms_per_step'])
b=l['bf16']; print('bf16', b['mode'], b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['whole_path']['frac'], b['concurrent_streams'])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['per_image']['batches'])
PY
