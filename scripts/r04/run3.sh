#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_3; mkdir -p $O
for b in bbx3lab_stamp bbx3lab_NOB_stamp bbx3lab_NOA_stamp bbx3lab_NOBAR_stamp bbx3lab_NOA_NOB_stamp bbx3lab_NOA_NOB_NOBAR_stamp; do echo "== $b"; timeout 120 ./popnet_amd/build/$b 32 112 112 20; done > $O/bbx3lab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16x3_fused" > $O/pytest_bbx3.log 2>&1; echo "rc $?" >> $O/pytest_bbx3.log
tail -n 4 $O/pytest_bbx3.log; cat $O/bbx3lab.log
