#!/bin/bash
# round-5 GPU check: usage check.sh <tag> [tests|bench|x3|train|all ...]
#   tests  the whole -m gpu suite + smoke          bench  the driver's command (N = 1, 20 steps)
#   x3     the bf16x3 pipelined region only         train  train-step timing (fp32 + bf16x3) + rocprofv3 kernel stats of the step
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift; O=gpurun_out/$tag; mkdir -p $O
for what in "$@"; do case $what in
tests)
  S=$(date +%s); timeout 3000 python -m pytest tests/ -x -q -m gpu ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; echo "pytest wall $(( $(date +%s) - S )) s"; tail -n 5 $O/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 1 $O/smoke.log ;;
bench)
  S=$(date +%s); timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver-cmd bench wall $(( $(date +%s) - S )) s"
  python3 scripts/r05/show_bench.py $O/bench_driver_cmd.json ;;
bench200)
  timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 scripts/r05/show_bench.py $O/bench_default.json ;;
quick)     # headline region only (no children, no CPU legs): A/B runs
  timeout 600 python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline ${BENCH_ARGS} > $O/quick_bf16.json 2> $O/quick_bf16.err; python3 scripts/r05/show_bench.py $O/quick_bf16.json ;;
x3)
  timeout 600 python3 bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --precision bf16x3 ${BENCH_ARGS} > $O/quick_x3.json 2> $O/quick_x3.err; python3 scripts/r05/show_bench.py $O/quick_x3.json ;;
yolo)
  timeout 600 python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --net yolo --no-h2d > $O/quick_yolo.json 2> $O/quick_yolo.err; python3 scripts/r05/show_bench.py $O/quick_yolo.json ;;
train)
  for p in fp32 bf16x3; do timeout 600 python3 scripts/train_bench.py 32 10 $p 2>&1 | grep -v amdgpu.ids | tee -a $O/train_bench.txt; done
  for p in fp32 bf16x3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_$p -o runc -- python3 scripts/train_bench.py 32 6 $p > $O/train_prof_$p.log 2>&1
    f=$(ls $O/train_$p/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/train_step_${p}_kernel_stats.csv
  done
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete ;;
esac; done
ls $O
