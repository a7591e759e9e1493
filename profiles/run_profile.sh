#!/bin/bash
# Reproduce the rocprofv3 evidence under profiles/ (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profile.sh r03 trace fetch write sq     # then, in a second call (the box does not persist):
#   bash profiles/run_profile.sh r03 cfetch cwrite csq
# trace: --kernel-trace --stats (per-kernel time). fetch / write: --pmc FETCH_SIZE / WRITE_SIZE, each in its own run without
# any trace option besides the kernel trace (MI355X_MICROARCH.md "HBM", PMC slots). sq: the SQ wave-lifetime split. The c*
# passes are the same counters for the tower convolution.
# Every pass runs the SAME command shape as the bench (real-net alignment, warm moves): the trees of a pass's timed window are
# then the bench's own, so counters and algorithmic bytes of the window describe one tree shape (round 2 aligned the PMC passes
# with the stub evaluator and compared counters at d-bar 1.6 with bytes at d-bar 2.4).
# Raw profiler output goes to /tmp (it exceeds what gpurun copies back); summaries are merged into profiles/ and copied to
# gpurun_out/profiles_TAG/. A heartbeat line per minute keeps the run visibly alive.
set -eo pipefail
TAG=${1:-r04}
shift || true
PASSES=${*:-trace fetch write sq cfetch cwrite csq}   # + cmfma, hfetch, hwrite (the evaluator's tail kernels) on request
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG
LOG=$ROOT/gpurun_out/prof_$TAG.log
mkdir -p "$OUT" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 60; echo "[heartbeat] $(date +%T) $(ls "$OUT" | tr '\n' ' ')" >> "$LOG"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null || true' EXIT
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
# Counter collection serialises every profiled dispatch (~5 ms each): a pass that profiled all ~225 k kernels of the bench's
# set-up did not finish in 20 minutes. So the counter passes filter by kernel (--kernel-include-regex): the k_step passes keep the
# bench's full command shape (what decides the tree shape) and profile the ~1.3 k k_step launches only; the convolution's counters
# do not depend on the trees and come from short passes (stub alignment, no warm moves).
KS="--kernel-include-regex k_step"
CV="--kernel-include-regex k_conv3x3"
SHORT="--steps 24 --warmup 2 --no-cpu-baseline --warm-moves 0 --align-evaluator stub"
pmc() { # pmc NAME "COUNTERS" "FILTER" bench-args...
  local name=$1 counters=$2 filter=$3; shift 3
  rocprofv3 --pmc $counters $filter --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o "$name" -- python3 "$ROOT/bench.py" "$@" > "$OUT/pmc_${name}_bench.json" 2> "$OUT/pmc_$name.err"
}
for P in $PASSES; do
  echo "$P pass starts $(date +%T)" >> "$LOG"
  case $P in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 120 --warmup 8 --no-cpu-baseline > "$OUT/trace_bench.json" 2> "$OUT/trace.err" ;;
    fetch) pmc fetch FETCH_SIZE "$KS" --steps 24 --warmup 2 --no-cpu-baseline ;;
    write) pmc write WRITE_SIZE "$KS" --steps 24 --warmup 2 --no-cpu-baseline ;;
    sq) pmc sq "$SQ" "$KS" --steps 24 --warmup 2 --no-cpu-baseline ;;
    cfetch) pmc cfetch FETCH_SIZE "$CV" $SHORT ;;
    cwrite) pmc cwrite WRITE_SIZE "$CV" $SHORT ;;
    csq) pmc csq "$SQ" "$CV" $SHORT ;;
    hfetch) pmc hfetch FETCH_SIZE "--kernel-include-regex k_head_conv1x1|k_fc_f16|k_fc_wide_f16|k_pack_live_planes|k_softmax_gather|k_cache_plan|k_cache_probe|k_value_out" $SHORT ;;
    hwrite) pmc hwrite WRITE_SIZE "--kernel-include-regex k_head_conv1x1|k_fc_f16|k_fc_wide_f16|k_pack_live_planes|k_softmax_gather|k_cache_plan|k_cache_probe|k_value_out" $SHORT ;;
    cmfma) pmc cmfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "$CV" $SHORT ;;
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
