#!/bin/bash
# Reproduce the rocprofv3 evidence under profiles/ (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profile.sh r03
# Pass 1: --kernel-trace --stats (per-kernel time). Passes 2/3: --pmc FETCH_SIZE / WRITE_SIZE, each in its
# own run without any trace option besides the kernel trace (MI355X_MICROARCH.md "HBM", PMC slots).
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 120 --warmup 8 --no-cpu-baseline > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace pass done"
# PMC passes: the SAME command shape as the bench (alignment with the real net: the trees of the timed window are then the
# bench's own trees, so the counters and the algorithmic bytes of the window describe one and the same tree shape; round 2
# aligned these passes with the stub evaluator and compared counters at d-bar 1.6 with bytes at d-bar 2.4)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
echo "write pass done"
# SQ wave-lifetime split (8 SQ slots in one pass): where k_step's and the convolution's waves spend their cycles
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 "$ROOT/bench.py" --steps 24 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq_bench.json" 2> "$OUT/pmc_sq.err"
echo "sq pass done"
python3 "$ROOT/profiles/summarize.py" "$OUT" "$TAG"
# gpurun merges at most 64 MiB back: keep the summaries and the small per-pass outputs, drop the raw per-dispatch traces
du -sh "$OUT" || true
find "$OUT" -type f -size +2M -delete || true
mkdir -p "$ROOT/gpurun_out/profiles_$TAG" && cp "$ROOT/profiles/${TAG}_summary.json" "$ROOT/profiles/${TAG}_kernel_stats.csv" "$ROOT/profiles/pmc_summary.json" "$ROOT/gpurun_out/profiles_$TAG/"
