# timing-only ablations of conv3_kernel (wrong results by construction) + timeline, current kernel
cd popnet_amd/build
export NBUF=1
for b in convlab convlab_NOA convlab_NOB convlab_NODMA convlab_NONE; do
  printf "%-14s level  " $b; GROUP="128:128,128:64" timeout 60 ./$b 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch"
  printf "%-14s 112    " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 0 | grep "us/launch"
done
GROUP="128:128,128:64" timeout 60 ./convlab_stamp 32 28 28 256 256 3 0 200 v3 0 | grep "us/launch\|stamps\|ends"
timeout 60 ./convlab_stamp 32 112 112 64 64 3 1 200 v3 1 | grep "us/launch\|stamps\|ends"
