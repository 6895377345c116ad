# three alternating repeats of auto / rect-512 / flat on the layers where one sweep (sweep_c8_tilings.sh) had another form
# ahead of the planner by more than the 2 % noise of a single run: all within noise when repeated (round 5)
for r in 1 2 3; do
  for L in dae.conv3_1 dae.up_conv3 dae.up_conv2 fcn.conv5_1; do
    for v in auto r512 flat; do
      unset IISEG_C8_TALL IISEG_C8_TILING
      case $v in r512) export IISEG_C8_TALL=1 IISEG_C8_TILING=1;; flat) export IISEG_C8_TILING=2;; esac
      python scripts/c8_layer.py $L 40 2>/dev/null | sed "s/^/$v /"
    done
  done
done
