cd popnet_amd/build
export NBUF=1
for r in 1 2; do for b in convlab convlab_db4; do
  printf "%-12s level  " $b; GROUP="128:128,128:64" timeout 60 ./$b 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch"
  printf "%-12s 112res " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 1 | grep "us/launch"
  printf "%-12s 112    " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 0 | grep "us/launch"
  printf "%-12s 56     " $b; timeout 60 ./$b 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch"
done; done
