#!/bin/bash
# per-workgroup timeline of k_tile_render (diagnostic build; on the GPU box):  profiles/timeline.sh [extra -D flags]
#   runs the persistent launch and, for comparison, a workgroup per tile (SVGR_TILE_WGS_PER_CU=0) -> gpurun_out/timeline.txt, timeline_one.txt
cd "$GRAFT_REPO_ROOT"
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc EXTRA="-DSVGR_DBG_TIMELINE $*" 2> gpurun_out/timeline_build.err || { echo BUILD FAILED; tail gpurun_out/timeline_build.err; exit 1; }
SVGR_DBG_TIMELINE=gpurun_out/timeline.bin timeout -k 10 120 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > /dev/null 2> gpurun_out/timeline_run.err
python profiles/timeline.py gpurun_out/timeline.bin > gpurun_out/timeline.txt; head -8 gpurun_out/timeline.txt
SVGR_TILE_WGS_PER_CU=0 SVGR_DBG_TIMELINE=gpurun_out/timeline_one.bin timeout -k 10 120 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > /dev/null 2>> gpurun_out/timeline_run.err
python profiles/timeline.py gpurun_out/timeline_one.bin > gpurun_out/timeline_one.txt; head -8 gpurun_out/timeline_one.txt
rm -f gpurun_out/timeline.bin gpurun_out/timeline_one.bin
make -s -C svgrasterize.py_amd/csrc clean && make -s -C svgrasterize.py_amd/csrc
