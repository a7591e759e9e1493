#!/bin/bash
# The one-game path after ccz_scouted_run (device-side loop of hit simulations), one box:
#  1. sweep over the number of scouts at n_playout = 200 (BASELINE configs[0]'s shape)      -> r06_single_board.json
#  2. n_playout = 1600 (the reference's default), 0 and 10 scouts                           -> r06_single_board_n1600.json
#  3. device loop against the host loop, interleaved A/B at both sizes + a 40-move game     -> r06_device_loop_ab.json
#  4. kernel-level timeline of the evaluator (rocprofv3 --kernel-trace)                     -> r06_single_board_timeline.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
SCOUTS=0,3,7,10,15,31 N_PLAYOUT=200 MOVES=8 python3 profiles/single_board_scouts.py > "$OUT/r06_single_board.json" 2> "$OUT/r06_single_board.err"
echo "sweep done"
SCOUTS=0,10 N_PLAYOUT=1600 MOVES=4 python3 profiles/single_board_scouts.py > "$OUT/r06_single_board_n1600.json" 2>> "$OUT/r06_single_board.err"
echo "n1600 done"
for rep in 1 2; do for l in 1 0; do
  CCZ_SCOUT_DEVICE_LOOP=$l SCOUTS=10 N_PLAYOUT=1600 MOVES=4 python3 profiles/single_board_scouts.py > "$OUT/ab_n1600_loop${l}_$rep.json" 2>> "$OUT/r06_single_board.err"
  CCZ_SCOUT_DEVICE_LOOP=$l SCOUTS=10 N_PLAYOUT=200 MOVES=8 python3 profiles/single_board_scouts.py > "$OUT/ab_n200_loop${l}_$rep.json" 2>> "$OUT/r06_single_board.err"
done; done
for l in 1 0; do CCZ_SCOUT_DEVICE_LOOP=$l SCOUTS=10 N_PLAYOUT=400 MOVES=40 python3 profiles/single_board_scouts.py > "$OUT/ab_game_loop$l.json" 2>> "$OUT/r06_single_board.err"; done
echo "ab done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sb
SCOUTS=10 N_PLAYOUT=200 MOVES=3 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -o t -- python3 "$ROOT/profiles/single_board_scouts.py" > /tmp/sb.json 2> /tmp/sb.err
python3 "$ROOT/profiles/small_timeline.py" "$(ls /tmp/sb/*kernel_trace.csv /tmp/sb/*/*kernel_trace.csv 2>/dev/null | head -1)" > "$OUT/r06_single_board_timeline.json"
echo "timeline done"
