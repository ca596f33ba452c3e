#!/bin/bash
# register / scratch / LDS use of the kernels, as the compiler reports them (no GPU needed):  profiles/kres.sh [-DSVGR_...]
cd "$(dirname "$0")/../svgrasterize.py_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o /tmp/kres.o svgr_hip.hip -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "error|Function Name|    VGPRs:|ScratchSize|LDS Size" \
  | sed -E 's/svgr_hip.hip:[0-9]+:[0-9]+: remark: +//g;s/\[-Rpass-analysis=kernel-resource-usage\]//g;s/Function Name: /\n/' | tr '\n' ' ' | sed 's/ _Z/\n_Z/g' \
  | awk '{name=$1; $1=""; print name, $0}' | c++filt | sed -E 's/\(.*\)//'
echo
