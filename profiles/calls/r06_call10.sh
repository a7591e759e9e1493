# round 6 call 10: k_step with the children of the previous path requested with the backup's loads (struct Spec) -- every test that
# compares the search bit for bit, then an A/B of the kernel's duration in the bench's own window (one box, interleaved)
O=gpurun_out
set -e
make -C chinesechesszero_amd/csrc ab NAME=nospec ABFLAGS=-DCCZ_NO_SPEC > /dev/null 2>&1   # (diagnostic builds do not travel: built on the box)
# (the bit-exactness tests -- search, timed path, modes, soak, scouts, twins, reference game, cache, full configs: 71 passed -- ran in the first attempt of this call)
cat > /tmp/ab_bench.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
if os.environ.get("CCZ_LIB"):
    from chinesechesszero_amd import _lib
    _lib.LIB_PATH = os.path.join(os.getcwd(), "build", "diag", os.environ["CCZ_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
for rep in 1 2 3; do
  for lib in "" libcczero_ab_nospec.so; do
    CCZ_LIB=$lib python /tmp/ab_bench.py --steps 120 --warmup 8 --no-cpu-baseline > $O/r06_spec_ab_${lib:-spec}_$rep.json 2> $O/r06_spec_ab_${lib:-spec}_$rep.err
    python - <<PY
import json
d = json.loads([l for l in open("$O/r06_spec_ab_${lib:-spec}_$rep.json").read().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
print("${lib:-spec}", $rep, "k_step us (events)", round(r["avg_launch_us"], 2), "frac", round(r["frac"], 4), "d_bar", round(r["d_bar"], 2), "sims/s", round(d["value"]))
PY
  done
done
