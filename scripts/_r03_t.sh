mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r03/build.log 2>&1 || { tail -20 gpurun_out/r03/build.log; exit 1; }
for dbg in 0 3 11 16 24; do
  for L in conv1_2 conv2_2 conv4_2 conv5_1y; do
    echo -n "dbg=$dbg " ; IISEG_BF16_DEBUG=$dbg ONLY=$L timeout -k 5 120 python scripts/bench_c8.py 2>/dev/null | grep "$L" | awk '{print $1,$2,$3,$4,$11,$12}'
  done
done > gpurun_out/r03/abl_c8_b.log 2>&1
cat gpurun_out/r03/abl_c8_b.log
