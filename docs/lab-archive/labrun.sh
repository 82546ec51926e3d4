# conv3_kernel timeline stamps and timing-only ablations (scripts/convlab.hip); run from the repo root on the GPU box.
# Binaries: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHAVE_CONV3 [-DPN_STAMP] [-DPN_CONV3_FAKE_{NOA,NOB,SAMEA,LINDMA,NOWRAP}] -Ipopnet_amd/csrc scripts/convlab.hip
cd popnet_amd/build
export NBUF=1
echo "== level: 256->256 + 128->128 + 128->64 (3x3, 28x28, B=32) in one launch, conv3_kernel<3,4,1,1>"
for b in convlab convlab_NOA convlab_SAMEA convlab_NOB convlab_NOAB convlab_LINDMA convlab_nowrap; do
  printf "%-16s " $b; GROUP="128:128,128:64" timeout 30 ./$b 32 28 28 256 256 3 0 50 v3 0 | grep "us/launch"
done
echo "== timeline of that launch (shader cycles; stamps: 0 start, 1 DMA issued, 2 first barrier passed, 3/4 5/6 7/8 9/10 chunk MFMAs done / hand-over done, 11 epilogue start, 12 end)"
GROUP="128:128,128:64" timeout 30 ./convlab_stamp 32 28 28 256 256 3 0 3 v3 0 | grep "stamps\|block starts"
echo "== single problems"
for s in "32 28 28 256 256 3 0" "32 28 28 128 128 3 0" "32 56 56 128 128 3 0" "32 112 112 64 64 3 1"; do
  for k in old v3; do cfg=$(echo $s | awk '{print $7}'); if [ $k = old ] && [ "$s" = "32 112 112 64 64 3 1" ]; then s2="32 112 112 64 64 3 4"; else s2="$s"; fi
  timeout 30 ./convlab $s2 50 $k 0 | grep "us/launch"; done
done
