#!/bin/bash
L=popnet_amd/build
for q in 3 6 9; do
  echo "== BQ $q"
  $L/conv4lab_bq$q 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:64 $L/conv4lab_bq$q 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=64:64 $L/conv4lab_bq$q 32 28 28 128 128 1000 | grep -v "mismatch"
  $L/conv4lab_bq$q 32 56 56 128 128 1000 | grep -v "mismatch"
done
