# round 6 call 7: scout + probe + plan as one launch -- tests, then the single-board profile again
O=gpurun_out
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_scouts.py tests/test_gpu_frontends_parity.py -x -q -m gpu > $O/r06_scouts_tests2.log 2>&1 || { tail -60 $O/r06_scouts_tests2.log; exit 1; }
tail -2 $O/r06_scouts_tests2.log
timeout -k 10 600 python profiles/single_board_scouts.py > $O/r06_single_board.json 2> $O/r06_single_board.err || { tail -20 $O/r06_single_board.err; exit 1; }
python - <<'PY'
import json; j = json.load(open("gpurun_out/r06_single_board.json")); print(j["same_moves_whatever_the_scouts"]); [print(r["scouts"], round(r["sims_per_sec"]), round(r["us_per_playout"], 1), r.get("evaluator_calls_per_simulation"), r.get("pieces_us")) for r in j["by_scouts"]]
PY
