#!/bin/bash
# Collect the rocprofv3 evidence of one round ON THE GPU BOX and leave only the summaries under gpurun_out/prof_<TAG>/
# (copy them into profiles/ afterwards):   gpurun -- 'bash tools/profile_round.sh r02a [workload]'
#   1. --kernel-trace --stats            -> <TAG>_<wl>_kernel_stats.csv + the bench line printed under the profiler
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only: MI355X_MICROARCH.md HBM section)
#   3. two --pmc SQ_* passes             -> <TAG>_<wl>_sq_counters.txt / .json (VALU per wave, LDS conflicts, clock)
# and kernel_counters.json (what bench.py reads for roofline.traffic / roofline.valu, stamped with the kernel source hash).
set -eo pipefail
TAG="$1"; WL="${2:-2160p}"
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$R/gpurun_out/prof_${TAG}"; RAW="/tmp/pqa_prof_${TAG}_${WL}"
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
cd /tmp
say() { echo "$(date +%T) $*" | tee -a "$OUT/progress.log"; }
# every pass keeps its stdout and its whole stderr (bench.py's stage markers and, with PQA_TRACE=1, the library's per-launch
# lines are in it) under $OUT/logs/: a pass that stalls is localised by its last marker, not thrown away
mkdir -p "$OUT/logs"
export PQA_BENCH_STAGES=1
keep() { cp "$RAW/$1.err" "$OUT/logs/${TAG}_${WL}_$1.stderr.txt" 2>/dev/null || true; [ -f "$RAW/$1.out" ] && cp "$RAW/$1.out" "$OUT/logs/${TAG}_${WL}_$1.stdout.txt" || true; }
# a pass that was KILLED at its time limit (124 / 137) ends the collection: no further GPU step after a timeout
failed() { rc=$?; say "the $1 counter pass failed or timed out: its figures will be missing; last lines of its stderr (all of it: logs/${TAG}_${WL}_$1.stderr.txt):"; tail -n 8 "$RAW/$1.err" | cut -c1-300 | tee -a "$OUT/progress.log"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then keep "$1"; say "pass $1 was killed at its time limit: stopping here"; exit 1; fi; }
T="timeout -k 10 ${PQA_PROF_TIMEOUT:-75}"
# 96 frames = one launch at 2160p (automatic batch 97), two of 48 at 2160p10 (frames per launch is then exact); three passes of
# the clip (1 warm-up + 2 steps), short, so a profiler stall costs little
FRAMES=96
B="$R/bench.py --workload $WL --steps 2 --warmup 1 --frames $FRAMES --no-cpu-baseline --no-other-configs --no-e2e"
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -- python3 $B > "$OUT/${TAG}_bench_${WL}_under_rocprof.json" 2> "$RAW/stats.err" || failed stats
keep stats
say "stats pass done"
$T rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$RAW/fetch" -- python3 $B > "$RAW/fetch.out" 2> "$RAW/fetch.err" || failed fetch
keep fetch
say "fetch pass done"
$T rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$RAW/write" -- python3 $B > "$RAW/write.out" 2> "$RAW/write.err" || failed write
keep write
say "traffic passes done"
$T rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$RAW/sqa" -- python3 $B > "$RAW/sqa.out" 2> "$RAW/sqa.err" || failed sqa
keep sqa
say "SQ pass A done"
$T rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d "$RAW/sqb" -- python3 $B > "$RAW/sqb.out" 2> "$RAW/sqb.err" || failed sqb
keep sqb
say "SQ passes done"
BATCH=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['config']['batch'])" "$OUT/${TAG}_bench_${WL}_under_rocprof.json")
# frames per launch averaged over all launches (the last batch of a 300-frame clip is partial)
FPL=$(python3 -c "import math; F=$FRAMES; B=$BATCH; print(F / math.ceil(F / B))")
python3 "$R/tools/sq_counters.py" "$RAW/sqa" "$RAW/sqb" --json="$RAW/sq.json" > "$OUT/${TAG}_${WL}_sq_counters.txt"
python3 "$R/tools/summarize_rocprof.py" --stats "$RAW/stats" --fetch "$RAW/fetch" --write "$RAW/write" --tag "${TAG}_${WL}" \
        --frames-per-launch "$FPL" --workload "$WL" --sq-json "$RAW/sq.json" --out "$OUT"
cp "$RAW/sq.json" "$OUT/${TAG}_${WL}_sq_counters.json"
rm -rf "$RAW"
ls -la "$OUT"
