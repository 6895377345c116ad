# every C8 layer class of configs[1] under each forced tiling (one box, 30 launches each)
set -e
out=$1
: > $out
for L in fcn.conv1_2 fcn.conv2_1 fcn.conv2_2 fcn.conv3_1 fcn.conv3_2 fcn.conv3_3 fcn.conv4_1 fcn.conv4_2 fcn.conv4_3 fcn.conv5_1 fcn.conv5_2 fcn.conv5_3 dae.conv2_1 dae.conv3_1 dae.conv4_1 dae.conv5_1y dae.conv6_1 dae.up_conv6p dae.up_conv5p dae.up_conv4 dae.up_conv3 dae.up_conv2; do
  for v in auto r256 r512 flat; do
    unset IISEG_C8_TALL IISEG_C8_TILING
    case $v in r256) export IISEG_C8_TALL=0 IISEG_C8_TILING=1;; r512) export IISEG_C8_TALL=1 IISEG_C8_TILING=1;; flat) export IISEG_C8_TILING=2;; esac
    python scripts/c8_layer.py $L 30 2>/dev/null | sed "s/^/$v /" >> $out
  done
done
cat $out
