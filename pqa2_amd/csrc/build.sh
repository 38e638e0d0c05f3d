#!/bin/bash
# Builds libpqa_vmaf.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result ${PQA_EXTRA_FLAGS:-}"
OBJS=()
PIDS=()
for f in vif vif_march vif_fixed adm adm_march adm_pyramid adm_fixed motion motion_march motion_fixed psnr_ssim luma_stats finalize ingest pqa_api; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ kernels.h -nt "$f.o" ] || [ pqa_device.h -nt "$f.o" ] || [ march_common.h -nt "$f.o" ] || [ ingest.h -nt "$f.o" ] || [ adm_chain.h -nt "$f.o" ] || [ host_pack.h -nt "$f.o" ] || [ host_ring.h -nt "$f.o" ] || [ ../../include/pqa_vmaf.h -nt "$f.o" ]; then
    rm -f "$f.o"   # a failed compile must not leave a stale object for the link step
    # adm_march: the SLP vectorizer pairs unrelated scalar multiplies of the decouple chain into v_pk_* ops at the price of
    # register moves and explicit abs (v_and) -- 4-clock instructions where 2-clock ones did (tools/ubench/valu_ops.hip)
    PERFILE=""; { [ "$f" = adm_march ] || [ "$f" = adm_pyramid ] || [ "$f" = motion_march ]; } && PERFILE="-fno-slp-vectorize"
    $HIPCC $FLAGS $PERFILE -c "$f.hip" -o "$f.o" &
    PIDS+=("$!")
  fi
  OBJS+=("$f.o")
done
# a bare `wait` returns 0 whatever the jobs did: check every compile's own status
for p in "${PIDS[@]:-}"; do
  [ -z "$p" ] && continue
  wait "$p" || { echo "compile failed (pid $p)" >&2; exit 1; }
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o libpqa_vmaf.so "${OBJS[@]}"
echo "built $(pwd)/libpqa_vmaf.so"
