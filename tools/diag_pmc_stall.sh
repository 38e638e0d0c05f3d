#!/bin/bash
# ONE counter pass of a workload under rocprofv3 --pmc with bench.py's stage markers and the library's per-launch trace
# (PQA_TRACE=1: synchronise after every launch and name it) on stderr, everything kept under gpurun_out/diag_<TAG>/.
#   gpurun -- 'bash tools/diag_pmc_stall.sh r03a 2160p10 [frames] [trace 0|1] [counter] [stages 0|1]'
# stages 0 = bench.py exactly as round 2 ran it (the clip generator then queues its ~7 000 torch launches without a sync)
# Made to localise the 2160p10 stall of round 2 from one run; never loops, never retries.
set -o pipefail
TAG="$1"; WL="${2:-2160p10}"; FRAMES="${3:-96}"; TRACE="${4:-1}"; CTR="${5:-FETCH_SIZE}"; STAGES="${6:-1}"
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$R/gpurun_out/diag_${TAG}"; RAW="/tmp/pqa_diag_${TAG}"
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp PQA_BENCH_STAGES="$STAGES" PQA_TRACE="$TRACE"
cd /tmp
NAME="${WL}_f${FRAMES}_trace${TRACE}_stages${STAGES}_${CTR}"
# stderr goes straight into gpurun_out/ (merged back even when the pass is killed)
timeout -k 10 "${PQA_DIAG_TIMEOUT:-100}" rocprofv3 --kernel-trace --output-format csv --pmc $CTR -d "$RAW/$NAME" -- \
  python3 "$R/bench.py" --workload "$WL" --steps 2 --warmup 1 --frames "$FRAMES" --no-cpu-baseline --no-other-configs --no-e2e \
  > "$OUT/$NAME.stdout.txt" 2> "$OUT/$NAME.stderr.txt"
rc=$?
echo "$(date +%T) $NAME rc=$rc" | tee -a "$OUT/progress.log"
grep -E "^\[(bench|pqa trace)" "$OUT/$NAME.stderr.txt" | tail -n 12
# dispatch count per kernel so far (the trace csv is only complete for a pass that ended by itself)
find "$RAW/$NAME" -name '*kernel_trace.csv' -exec sh -c 'cut -d, -f8- "$1" | sort | uniq -c | sort -rn | head -40' _ {} \; > "$OUT/$NAME.dispatch_counts.txt" 2>/dev/null
rm -rf "$RAW"
exit $rc
