mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_n.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_n.log
timeout -k 10 900 python bench.py > gpurun_out/r03/bench_full_1.json 2> gpurun_out/r03/bench_full_1.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_full_1.json'))
print('fp32', l['value'], l['ms_per_step'], 'roofline', l['roofline']['kernel'], l['roofline']['frac'], 'whole', l['roofline']['whole_path'])
for k in ('per_batch_only','full_recompute','strict_f64'): print(k, l[k]['value'], l[k]['ms_per_step'])
b=l['bf16']; print('bf16', b['mode'], b['value'], b['ms_per_step'], b['roofline']['kernel'], b['roofline']['frac'], b['roofline']['whole_path']['frac'])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['cores'], l['cpu_baseline']['per_image'], l['cpu_baseline']['batched'])
PY
