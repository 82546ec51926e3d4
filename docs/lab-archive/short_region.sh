# the same build with the driver's short regions (--steps 20 --warmup 5) and with the default 200-step regions, alternating on one box
one() { timeout 300 python3 bench.py --no-extras --no-cpu-baseline --no-h2d --reps 5 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['steps'], d['value'], d['value_stat']['runs'], d['roofline']['frac'])"; }
for i in 1 2; do one --steps 20 --warmup 5; one; done
one --steps 20 --warmup 5 --pipeline 2
one --steps 20 --warmup 5 --pipeline 4
