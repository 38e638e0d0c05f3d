#!/bin/bash
# Builds libpqa_vmaf.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
OBJS=()
for f in vif vif_fixed adm adm_fixed motion motion_fixed psnr_ssim luma_stats finalize pqa_api; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ kernels.h -nt "$f.o" ] || [ pqa_device.h -nt "$f.o" ] || [ ../../include/pqa_vmaf.h -nt "$f.o" ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
  fi
  OBJS+=("$f.o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o libpqa_vmaf.so "${OBJS[@]}"
echo "built $(pwd)/libpqa_vmaf.so"
