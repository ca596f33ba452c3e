#!/bin/bash
# Instruction mix of the geometry kernels (one PMC pass): profiles/pmc_geometry.sh   -> gpurun_out/pmc_geometry.txt
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcg
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcg -o c -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
for k in k_edge_emit k_flatten k_band_entries k_edge_count k_path_bbox; do
  python3 profiles/pmc_summary.py --only=$k gpurun_out/pmcg/c_counter_collection.csv
done | tee gpurun_out/pmc_geometry.txt
