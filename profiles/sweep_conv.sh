#!/bin/bash
# blur row-pass variants: profiles/sweep_conv.sh "rb8:-DSVGR_CONV_RB=8" "rb16:-DSVGR_CONV_RB=16"   -> gpurun_out/sweep_conv.txt
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep_conv.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  echo "== $name" >> $out
  timeout -k 10 120 python profiles/bench_layer_ops.py 2>/dev/null | grep "convolve\|convert" >> $out
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
