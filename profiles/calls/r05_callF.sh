# round 5: (1) what one GPU can show of the exchange on RCCL: the full-size three-move window with and without the N>1 path in a group of one,
# interleaved; (2) soaks of the asynchronous exchange over ~100 move boundaries with 2 and 4 ranks sharing the GPU (gloo), one of them with a slow rank
O=gpurun_out
set -e
for i in 1 2; do
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r05_g1_plain_$i.json 2> $O/r05_f1.err; echo "plain $i"
python bench.py --steps 1200 --warmup 16 --no-cpu-baseline --rccl-group-of-one > $O/r05_g1_rccl_$i.json 2> $O/r05_f2.err; echo "rccl $i"
done
SOAK="--backend gloo --share-gpu --boards 512 --blocks 2 --channels 256 --playout 32 --max-plies 40 --preroll-plies 40 --steps 3200 --warmup 4 --gather-plies 4096"
python bench.py --gpus 2 $SOAK > $O/r05_soak_2ranks.json 2> $O/r05_f3.err; echo "soak 2"
python bench.py --gpus 4 $SOAK > $O/r05_soak_4ranks.json 2> $O/r05_f4.err; echo "soak 4"
python bench.py --gpus 2 $SOAK --slow-rank 1:0.05 > $O/r05_soak_2ranks_slow.json 2> $O/r05_f5.err; echo "soak 2 slow"
python - <<'PY'
import json
def L(f): return json.loads([l for l in open(f"gpurun_out/{f}").read().splitlines() if l.startswith("{")][-1])
for f in ("r05_g1_plain_1", "r05_g1_rccl_1", "r05_g1_plain_2", "r05_g1_rccl_2"):
    d = L(f + ".json"); m = d.get("multi_gpu", {})
    print(f, round(d["value"]), round(d["ms_per_step"], 3), {k: m.get(k) for k in ("collectives_in_window", "collectives_in_drain", "exchanges_decided", "exchange_host_ms_rank0", "exchange_max_call_ms_per_rank", "bytes_sent_rank0", "rows_gathered", "drain_ms_rank0", "backend")})
for f in ("r05_soak_2ranks", "r05_soak_4ranks", "r05_soak_2ranks_slow"):
    d = L(f + ".json"); m = d["multi_gpu"]
    print(f, round(d["value"]), round(d["ms_per_step"], 3), d["move_boundary"]["in_window"], {k: m.get(k) for k in ("rank_step_ms", "collectives_in_window", "collectives_in_drain", "exchanges_decided", "exchanges_without_a_collective", "exchange_host_ms_rank0", "exchange_max_call_ms_per_rank", "rows_gathered", "games_gathered", "replay_rows_total", "bad_records", "error_flags_any", "backlog_peak_plies_rank0")})
PY
