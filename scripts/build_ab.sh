#!/bin/bash
# A/B library: libiiseg_hip.so with csrc/<file> taken from a git revision (everything else as built now).
# Usage: bash scripts/build_ab.sh <rev> <file.hip> <out.so>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rev=$1; f=$2; out=$3
tmp=$(mktemp -d)
git -C $R show $rev:iterative_inference_segm_amd/csrc/$f > $tmp/$f
for h in common.h conv_common.h c8_common.h; do git -C $R show $rev:iterative_inference_segm_amd/csrc/$h > $tmp/$h 2>/dev/null || cp $R/iterative_inference_segm_amd/csrc/$h $tmp/$h; done
# (per-source flags as in iterative_inference_segm_amd/build.py EXTRA_FLAGS)
extra=$(python3 -c "import sys; sys.path.insert(0, '$R'); from iterative_inference_segm_amd.build import EXTRA_FLAGS; print(' '.join(EXTRA_FLAGS.get('$f', [])))")
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $extra -I$R/include -I$tmp -c $tmp/$f -o $tmp/ab.o
objs=""
for o in $R/iterative_inference_segm_amd/build/*.o; do
  if [ "$(basename $o)" = "${f%.hip}.o" ]; then objs="$objs $tmp/ab.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
rm -rf $tmp
echo built $out
