set -e
O=gpurun_out
python bench.py --boards 1024 --steps 400 --warmup 16 --no-cpu-baseline > $O/r04_bench_cfg2_1024boards_final.json 2> $O/r04_m1.err; echo cfg2
python profiles/single_board.py > $O/r04_single_board_final.json 2> $O/r04_m2.err; echo single
python bench.py --gpus 2 --backend gloo --share-gpu --boards 1024 --steps 40 --warmup 4 > $O/r04_rehearsal_2ranks_gloo_final.json 2> $O/r04_m3.err; echo rehearsal
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_cfg2_1024boards_final.json")); print("1024 boards", round(d["value"]), d["ms_per_step"])
d = json.load(open("gpurun_out/r04_single_board_final.json")); print("single", d["mcts_ai_get_action"])
d = json.loads(open("gpurun_out/r04_rehearsal_2ranks_gloo_final.json").read().strip().splitlines()[-1]); print("2 ranks", round(d["value"]), d["multi_gpu"].get("ranks_seen"), d["error_flags_any"])
PY
