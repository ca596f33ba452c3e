#!/bin/bash
# per-phase lifetimes of k_path_build's workgroups (s_memrealtime stamps): [PB_WORKLOAD=tiger2048] profiles/pb_stamp.sh   (on the GPU box)
cd "$GRAFT_REPO_ROOT"
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc EXTRA="-DSVGR_DBG_PB_STAMP $*" 2> gpurun_out/pb_stamp_build.err || { echo BUILD FAILED; tail gpurun_out/pb_stamp_build.err; exit 1; }
SVGR_DBG_PB_DUMP=gpurun_out/pb_stamp.bin timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 ${PB_WORKLOAD:+--workload $PB_WORKLOAD} 2>&1 | grep "pb stamp"
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
