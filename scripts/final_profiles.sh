# round-end evidence: bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 600 gpurun_out/$tag/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o runc -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/$tag/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats_eager1 -o runc -- python3 bench.py --no-cpu-baseline --no-extras --no-h2d --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5 > gpurun_out/$tag/stats_eager1.log 2>&1
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_$set -o runc -- python3 scripts/fwd_only.py 3 > gpurun_out/$tag/pmc_$set.log 2>&1
done
python3 scripts/make_traffic_json.py gpurun_out/$tag gpurun_out/$tag/conv_hbm_traffic.json | tail -30
