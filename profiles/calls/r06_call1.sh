# round 6 call 1: persistent tower kernel -- bit identity, then one layer in isolation (interleaved, one box)
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "persistent or group_of_16" > gpurun_out/r06_pers_tests.log 2>&1 || { tail -30 gpurun_out/r06_pers_tests.log; exit 1; }
tail -3 gpurun_out/r06_pers_tests.log
P256=$((65 + 256 + (256 << 16))); P128=$((65 + 256 + (128 << 16))); P248=$((65 + 256 + (248 << 16)))
for B in 4096 3712 2048 1024; do
  timeout -k 10 200 python profiles/conv_ab.py libcczero.so:65 libcczero.so:193 libcczero.so:$P256 libcczero.so:$P128 libcczero.so:$P248 --boards $B --rounds 7 --iters 10 --res 1 --relu-input 1 > gpurun_out/r06_pers_ab_$B.json 2>&1
  python - <<PY
import json; j=json.load(open("gpurun_out/r06_pers_ab_$B.json")); print($B, j["max_diff_vs_first"], {k: round(v["median"],1) for k,v in j["us"].items()})
PY
done
