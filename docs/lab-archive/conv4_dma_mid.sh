L=popnet_amd/build
for v in "" _PN4_DMA_MID "" _PN4_DMA_MID; do
  echo "== conv4lab$v"
  $L/conv4lab$v 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:64 $L/conv4lab$v 32 28 28 256 256 1000 | grep -v "mismatch"
  GROUP=128:128,128:128 $L/conv4lab$v 32 28 28 128 256 1000 | grep -v "mismatch"
  $L/conv4lab$v 32 56 56 128 128 1000 1 | grep -v "mismatch"
done
