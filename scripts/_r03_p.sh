mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
for geo in "512 512 52 64 39" "256 512 52 64 35" "512 1024 26 64 22" "256 256 105 64 64"; do
 for v in "IISEG_WBF_MINW=2" "IISEG_WBF_MINW=1" "IISEG_WBF_MINW=2 IISEG_BF16_FUSED_TILE=64" "IISEG_WBF_MINW=1 IISEG_BF16_FUSED_TILE=64"; do
  echo -n "$geo | $v | "; env $v IISEG_BF16_FORM=wino timeout -k 5 100 python scripts/time_layer.py $(echo $geo | cut -d' ' -f1-3) bf16 $(echo $geo | cut -d' ' -f4-5) 2>/dev/null | tail -1
 done
done > gpurun_out/r03/wbf_ab.log 2>&1
cat gpurun_out/r03/wbf_ab.log
