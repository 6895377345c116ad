mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
IISEG_DIST_BACKEND=gloo IISEG_FORCE_DEVICE=0 timeout -k 10 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_2rank_gloo.json 2> gpurun_out/r03/bench_2rank_gloo.err; echo "rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r03/bench_2rank_gloo.json'))
print({k:l[k] for k in ('value','n_gpus','ms_per_step','per_rank_images_per_s','metric_all_reduce_ms','scaling')}, [k for k in l if k in ('bf16','strict_f64','full_recompute')])
PY
