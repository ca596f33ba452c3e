#!/bin/bash
# build variants against the 8-rank emulation of config 4 (and the single GPU):  profiles/sweep_emulate.sh "name:-DSVGR_..." ...  -> gpurun_out/sweep_emulate.txt
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sweep_emulate.txt
: > $out
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean
  if ! make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2> gpurun_out/sweep_build_$name.err; then echo "$name BUILD FAILED" >> $out; continue; fi
  echo "== $name  ($flags)" >> $out
  for w in 1 8; do
    timeout -k 10 300 python3 profiles/emulate_rank.py --world $w --all --steps 60 --workload synth8192 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  world', d['world'], 'slowest', d['slowest']['ms_step'], [r['ms_step'] for r in d['ranks']])" >> $out
  done
done
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
cat $out
