#!/bin/bash
# conv4_kernel vs conv3_kernel on the stage-level shapes (run from the repo root on the GPU box)
L=popnet_amd/build
for shape in "32 28 28 256 256" "32 28 28 128 128" "32 28 28 192 256" "32 56 56 128 128" "3 28 28 128 128" "1 30 26 128 64" "32 28 28 128 256"; do
  $L/conv4lab $shape 200 2>&1 | grep -v mismatch
  $L/conv3lab $shape 200 2>&1 | grep -v mismatch
done
echo "--- residual"
$L/conv4lab 32 56 56 128 128 200 1 | grep -v mismatch
echo "--- levels"
GROUP=128:128,128:64 $L/conv4lab 32 28 28 256 256 200 | grep -v mismatch
GROUP=128:128,128:64 $L/conv3lab 32 28 28 256 256 200 | grep -v mismatch
GROUP=192:128,192:128 $L/conv4lab 32 28 28 192 256 200 | grep -v mismatch
GROUP=192:128,192:128 $L/conv3lab 32 28 28 192 256 200 | grep -v mismatch
GROUP=64:64 $L/conv4lab 32 28 28 128 128 200 | grep -v mismatch
GROUP=64:64 $L/conv3lab 32 28 28 128 128 200 | grep -v mismatch
echo "--- stamps"
$L/conv4lab_stamp 32 28 28 256 256 2000 | grep -v mismatch
GROUP=128:128,128:64 $L/conv4lab_stamp 32 28 28 256 256 2000 | grep -v mismatch
