# Same box, interleaved: the head convolutions in the last tower layer's epilogue (default) against a pass of their own (CCZ_FUSED_LAST=0)
set -e
O=gpurun_out
for i in 1 2 3; do
  python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_ab_fused_last_on_$i.json 2> $O/r04_fl_on_$i.err; echo "on $i done"
  CCZ_FUSED_LAST=0 python bench.py --steps 1200 --warmup 16 --no-cpu-baseline > $O/r04_ab_fused_last_off_$i.json 2> $O/r04_fl_off_$i.err; echo "off $i done"
done
python - <<'PY'
import json, glob
for k in ("on", "off"):
    v = [json.load(open(f))["value"] for f in sorted(glob.glob(f"gpurun_out/r04_ab_fused_last_{k}_*.json"))]
    print(k, [round(x) for x in v], round(sum(v) / len(v)))
PY
