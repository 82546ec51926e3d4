# A/B of one environment switch on the YoloPoseNet leg: ab_switch_yolo.sh SWITCH=1 [rounds]
SW="$1"; N="${2:-2}"
one() { timeout 300 python3 bench.py --net yolo --no-extras --no-cpu-baseline --no-h2d --reps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['value'], d['value_stat']['runs'], d['roofline']['conv_stack']['ms_per_step'], d['roofline']['conv_stack']['launches_per_step'])" "$1"; }
for i in $(seq $N); do
  env "$SW" bash -c "$(declare -f one); one '$SW'"
  one default
done
