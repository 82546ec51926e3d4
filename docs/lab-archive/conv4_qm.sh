#!/bin/bash
# quarter-major (conflict-free) vs half-major B images in conv4_kernel: conv4lab_qm vs conv4lab_bq6 (previous build)
L=popnet_amd/build
for b in conv4lab_bq6 conv4lab_qm; do
  echo "== $b"
  $L/$b 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:64 $L/$b 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=192:128,192:128 $L/$b 32 28 28 192 256 1000 | grep -v "mismatch"
  $L/$b 32 56 56 128 128 1000 | grep -v "mismatch"
done
