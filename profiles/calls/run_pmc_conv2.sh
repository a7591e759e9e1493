#!/bin/bash
# SQ counters of the two forms of the tower convolution on the same data (one process, interleaved): where do the waves wait?
# needs build/diag/libcczero_experiments.so: make -C chinesechesszero_amd/csrc experiments
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_conv2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT -o sq -- python3 $ROOT/profiles/conv_ab.py libcczero.so:1 libcczero_experiments.so:5 --boards 4096 --rounds 2 --iters 3 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for name in ("k_conv3x3_c256", "k_conv3x3_v2"):
        if name in k:
            a = acc[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    v = {c: x[0] / x[1] for c, x in d.items()}
    wc = v["SQ_WAVE_CYCLES"]
    print(k, {c: round(x) for c, x in v.items()}, "launches", list(d.values())[0][1])
    print("   wait_any %.3f wait_inst %.3f active %.3f lds_conflict/lds_active %.4f" % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
PY
find $OUT -type f -size +2M -delete || true
