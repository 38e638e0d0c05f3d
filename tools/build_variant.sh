#!/bin/bash
# Build a second libpqa_vmaf.so for in-process A/B timing (tools/ab_bench.py):
#   tools/build_variant.sh NAME [GIT_REF|work] [extra hipcc flags...]
# copies pqa2_amd/csrc + include from GIT_REF (default: work = the working tree) into build/variants/NAME/ and builds
# there.  build/ is git-ignored but travels to the GPU box with gpurun.  Prints the path of the library.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; REF="${2:-work}"; shift; shift || true
DST="$ROOT/build/variants/$NAME"
rm -rf "$DST"; mkdir -p "$DST"
if [ "$REF" = "work" ]; then
  mkdir -p "$DST/pqa2_amd" && cp -r "$ROOT/pqa2_amd/csrc" "$DST/pqa2_amd/csrc" && cp -r "$ROOT/include" "$DST/include"
  rm -f "$DST"/pqa2_amd/csrc/*.o "$DST"/pqa2_amd/csrc/*.so
else
  git -C "$ROOT" archive "$REF" pqa2_amd/csrc include | tar -x -C "$DST"
fi
PQA_EXTRA_FLAGS="$*" bash "$DST/pqa2_amd/csrc/build.sh" >/dev/null 2>"$DST/build.err" || { cat "$DST/build.err" >&2; exit 1; }
echo "$DST/pqa2_amd/csrc/libpqa_vmaf.so"
