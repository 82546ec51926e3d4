#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_13; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q > $O/pytest_train.log 2>&1; echo "rc $?" >> $O/pytest_train.log; tail -n 6 $O/pytest_train.log
timeout 600 python scripts/train_bench.py 32 20 fp32 2>&1 | grep -v amdgpu | tail -3
timeout 600 python scripts/train_bench.py 32 20 bf16x3 2>&1 | grep -v amdgpu | tail -3
