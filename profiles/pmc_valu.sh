#!/bin/bash
# VALU / SALU / LDS instruction counts of the tile kernel for build variants: profiles/pmc_valu.sh "name:flags" ...
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc EXTRA="$flags" 2>/dev/null
  rm -rf gpurun_out/pmcv_$name
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcv_$name -o c -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  echo "== $name"; python3 profiles/pmc_summary.py --only=k_tile_render gpurun_out/pmcv_$name/c_counter_collection.csv | grep SQ_
  python3 bench.py --no-cpu-baseline --steps 100 | python3 -c "import json,sys; d=json.load(sys.stdin); print('   tile ms', d['roofline']['avg_launch_ms'])"
done
make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc 2>/dev/null
