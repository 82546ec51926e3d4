#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_6; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity.py -x -q > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log
tail -n 5 $O/pytest_a.log
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_targets.py -x -q > $O/pytest_b.log 2>&1; echo "rc $?" >> $O/pytest_b.log
tail -n 5 $O/pytest_b.log
