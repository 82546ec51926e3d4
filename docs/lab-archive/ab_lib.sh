# A/B of an alternative library build on one box: ab_lib.sh path/to/libpopnet_variant.so [rounds]
LIBV="$1"; N="${2:-3}"
one() { timeout 300 python3 bench.py --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['value'], d['value_stat']['runs'], d['roofline']['conv_stack']['ms_per_step'], d['roofline']['avg_launch_us'])" "$1"; }
for i in $(seq $N); do
  POPNET_LIB_PATH="$LIBV" bash -c "$(declare -f one); one variant"
  one default
done
