#!/bin/bash
# round-3 conv4_kernel variants (scripts/conv4lab.hip built with the -D switches named in the binary): lone 256->256 level,
# the two imbalanced stage levels, the balanced first stage level and the 56x56 128->128 level
L=popnet_amd/build
for v in "" _PN4_LGKM3 _PN4_PITCH36 _PN4_PITCH36_PN4_LGKM3 _PN4_RING4 _PN4_WT_STORE _PN4_PITCH36_PN4_LGKM3_PN4_WT_STORE _PN4_RING4_PN4_WT_STORE; do
  echo "== conv4lab$v"
  $L/conv4lab$v 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:64 $L/conv4lab$v 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,64:64 $L/conv4lab$v 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:128 $L/conv4lab$v 32 28 28 128 256 1000 | grep -v "mismatch"
  $L/conv4lab$v 32 56 56 128 128 1000 1 | grep -v "mismatch"
done
