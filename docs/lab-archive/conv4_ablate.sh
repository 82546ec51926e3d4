#!/bin/bash
# timing-only ablations of conv4_kernel (wrong results by construction): where a k-step's cycles go
L=popnet_amd/build
for v in stamp NODMA_A NODMA_B NODMA_A_NODMA_B NOBAR NOLDS NODMA_A_NODMA_B_NOBAR NODMA_A_NODMA_B_NOBAR_NOLDS; do
  echo "== $v"
  $L/conv4lab_$v 32 28 28 256 256 1000 | grep -v "mismatch\|check"
  GROUP=128:128,128:64 $L/conv4lab_$v 32 28 28 256 256 1000 | grep -v "mismatch\|check"
done
