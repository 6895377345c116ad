set -x
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --no-full-recompute --no-strict-f64 --steps 8 > gpurun_out/r03/bench_c8_a.json 2> gpurun_out/r03/bench_c8_a.err; echo "rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_c8_a.json'))
print('fp32', l['value'], l['ms_per_step'])
b=l['bf16']; print('bf16', b['mode'], b['value'], b['ms_per_step'], b.get('delta_miou_vs_f32'))
r=b['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['per_kernel_ms_per_step'], r['per_kernel_tflops'], r['all_conv_ms_per_step'])
PY
