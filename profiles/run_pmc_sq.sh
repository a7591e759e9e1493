set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o sq -- python3 $ROOT/profiles/sim_microbench.py 4096 400 1 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for name in ("k_step","k_select","k_expand_backup","k_legal_moves","k_finish_move"):
        if name in k:
            a = acc[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, {c: round(v[0]/v[1]) for c, v in d.items()}, "launches", list(d.values())[0][1])
PY
