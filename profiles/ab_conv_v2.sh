# in-workload A/B of the two forms of the tower kernel (CCZ_CONV_V2) on ONE box; the shipped library ignores the bit:
# build with `make -C chinesechesszero_amd/csrc HIPFLAGS+=-DCCZ_CONV2` first (and rebuild without it afterwards)
B=${1:-4096}
run() { echo -n "boards $B $1: "; env $1 python bench.py --boards $B --steps 40 --warmup 5 --no-cpu-baseline --preroll-plies 8 --no-align 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['ms_per_step'],3), round(j['net_roofline']['avg_launch_us'],1))"; }
for rep in 1 2; do
run "CCZ_CONV_V2=0"
run "CCZ_CONV_V2=1"
run "CCZ_CONV_V2=0 CCZ_TOWER_CHAINS=1"
run "CCZ_CONV_V2=1 CCZ_TOWER_CHAINS=1"
done
