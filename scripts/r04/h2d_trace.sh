#!/bin/bash
# PCIe hand-over timeline: kernel + memory-copy trace of a short h2d-inclusive bench run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_h2d; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o run -- python3 bench.py --steps 40 --warmup 6 --reps 1 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ls -la $O/trace/*/ 2>/dev/null | head; python3 scripts/r04/h2d_analyze.py $O/trace | tee $O/analysis.txt
