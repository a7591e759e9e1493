#!/bin/bash
# Reproduce the rocprofv3 evidence under profiles/ (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profile.sh r03 trace fetch        # then, in a second call (the box does not persist):
#   bash profiles/run_profile.sh r03 write sq
# trace: --kernel-trace --stats (per-kernel time). fetch / write: --pmc FETCH_SIZE / WRITE_SIZE, each in its own run without
# any trace option besides the kernel trace (MI355X_MICROARCH.md "HBM", PMC slots). sq: the SQ wave-lifetime split.
# Every pass runs the SAME command shape as the bench (real-net alignment, warm moves): the trees of a pass's timed window are
# then the bench's own, so counters and algorithmic bytes of the window describe one tree shape (round 2 aligned the PMC passes
# with the stub evaluator and compared counters at d-bar 1.6 with bytes at d-bar 2.4).
# Raw profiler output goes to /tmp (it exceeds what gpurun copies back); summaries are merged into profiles/ and copied to
# gpurun_out/profiles_TAG/. A heartbeat line per minute keeps the run visibly alive.
set -eo pipefail
TAG=${1:-r03}
shift || true
PASSES=${*:-trace fetch write sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG
LOG=$ROOT/gpurun_out/prof_$TAG.log
mkdir -p "$OUT" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 60; echo "[heartbeat] $(date +%T) $(ls "$OUT" | tr '\n' ' ')" >> "$LOG"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null || true' EXIT
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for P in $PASSES; do
  echo "$P pass starts $(date +%T)" >> "$LOG"
  case $P in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 120 --warmup 8 --no-cpu-baseline > "$OUT/trace_bench.json" 2> "$OUT/trace.err" ;;
    fetch) rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err" ;;
    write) rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err" ;;
    sq) rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq_bench.json" 2> "$OUT/pmc_sq.err" ;;
    *) echo "unknown pass $P" >> "$LOG"; exit 2 ;;
  esac
  echo "$P pass done $(date +%T)" >> "$LOG"
done
python3 "$ROOT/profiles/summarize.py" "$OUT" "$TAG" >> "$LOG" 2>&1
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
cp "$ROOT/profiles/${TAG}_summary.json" "$ROOT/profiles/pmc_summary.json" "$ROOT/gpurun_out/profiles_$TAG/"
[ -f "$ROOT/profiles/${TAG}_kernel_stats.csv" ] && cp "$ROOT/profiles/${TAG}_kernel_stats.csv" "$ROOT/gpurun_out/profiles_$TAG/"
cp "$OUT"/*.err "$OUT"/*_bench.json "$ROOT/gpurun_out/profiles_$TAG/" 2>/dev/null || true
echo "all done $(date +%T)" >> "$LOG"
