#!/bin/bash
# Builds tuning variants of the engine library into apus_amd/variants/ (git-ignored, shipped by gpurun):
#   tools/build_variants.sh "name:-DFLAG=.. -DFLAG2=.." ...
# Select one at run time with APUS_GPU_LIB=apus_amd/variants/libapus_gpu_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p apus_amd/variants
for c in apus_amd/host/apus_*.c; do gcc -O2 -fPIC -std=gnu11 -Iinclude -c $c -o apus_amd/variants/$(basename ${c%.c}).o; done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c apus_amd/csrc/apus_engine.hip -o apus_amd/variants/eng_$name.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Name: _Z1[0-9]k_replica|VGPRs:|SGPRs Spill|Occupancy" | grep -A3 k_replica | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | tr '\n' ' '
  echo " <- $name"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o apus_amd/variants/libapus_gpu_$name.so apus_amd/variants/eng_$name.o apus_amd/variants/apus_*.o -lpthread
done
