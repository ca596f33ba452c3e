#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in v_f; do
  cp profiles/_ab/$v.hip svgrasterize.py_amd/csrc/svgr_hip.hip
  make -s -C svgrasterize.py_amd/csrc clean; make -s -C svgrasterize.py_amd/csrc 2>/dev/null
  for i in 1 2 3; do python3 bench.py --no-cpu-baseline | python3 -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ab_$v -o c -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  python3 profiles/pmc_summary.py --only=k_tile_render gpurun_out/pmc_ab_$v/c_counter_collection.csv
done
