#!/bin/bash
# One command for whoever has an ffmpeg built with libvmaf: runs the reference's three filter graphs on the committed
# golden clips and diffs everything against this repository's restatements (add --gpu on an MI355X box for the kernels).
#   tools/pin_with_ffmpeg.sh [--gpu]
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
FF="${FFMPEG:-ffmpeg}"
T="$(mktemp -d)"
rc=0
for clip in c64x48_8 c352x288_8 c200x120_10; do
  ref="$R/tests/golden/clips/${clip}_ref.y4m"; dis="$R/tests/golden/clips/${clip}_dist.y4m"
  for model in vmaf_v0.6.1 vmaf_float_v0.6.1; do
    "$FF" -hide_banner -loglevel error -i "$dis" -i "$ref" -lavfi "libvmaf=log_fmt=json:log_path=$T/$clip.$model.json:model=version=$model:n_threads=4" -f null -
    python3 "$R/tools/compare_libvmaf_log.py" "$T/$clip.$model.json" "$ref" "$dis" "$@" || rc=1
  done
  "$FF" -hide_banner -loglevel error -i "$dis" -i "$ref" -lavfi "psnr=stats_file=$T/$clip.psnr.txt" -f null -
  "$FF" -hide_banner -loglevel error -i "$dis" -i "$ref" -lavfi "ssim=stats_file=$T/$clip.ssim.txt" -f null -
  python3 "$R/tools/compare_ffmpeg_stats.py" --psnr "$T/$clip.psnr.txt" --ssim "$T/$clip.ssim.txt" "$ref" "$dis" "$@" || rc=1
done
echo "outputs kept in $T"
exit $rc
