#!/bin/bash
# How full do the launch chains keep the chip at 1024 / 2048 / 4096 boards? One kernel trace each (the CSVs stay in /tmp on the box:
# 58 MB apiece), reduced by profiles/tower_timeline.py; output gpurun_out/r06_tower_timeline_<boards>.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for b in 1024 2048 4096; do
  rm -rf /tmp/tl_$b
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$b -o t -- python3 "$ROOT/bench.py" --boards $b --steps 60 --warmup 16 --no-cpu-baseline > /tmp/tl_${b}_bench.json 2> "$OUT/r06_tower_timeline_$b.err"
  python3 "$ROOT/profiles/tower_timeline.py" "$(ls /tmp/tl_$b/*kernel_trace.csv /tmp/tl_$b/*/*kernel_trace.csv 2>/dev/null | head -1)" /tmp/tl_${b}_bench.json $([ $b = 4096 ] && echo 400 || echo 120) > "$OUT/r06_tower_timeline_$b.json"
  echo "done $b"
done
