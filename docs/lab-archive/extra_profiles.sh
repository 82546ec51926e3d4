# rocprofv3 kernel stats of the two secondary legs of the bench line, eager on one stream (alone-times): bf16x3 parity mode, YoloPoseNet
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/extra
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/x3 -o runc -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-extras --no-h2d --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5 > gpurun_out/extra/x3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/yolo -o runc -- python3 bench.py --net yolo --no-cpu-baseline --no-extras --no-h2d --reps 1 --no-graph --pipeline 1 --steps 20 --warmup 5 > gpurun_out/extra/yolo.log 2>&1
ls gpurun_out/extra/x3 gpurun_out/extra/yolo
