cd popnet_amd/build
export NBUF=1
echo "== PT=7 reference"
GROUP="128:128,128:64" timeout 60 ./convlab 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch\|check"
timeout 60 ./convlab 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch"
timeout 60 ./convlab 32 56 56 64 128 3 0 2000 v3 0 | grep "us/launch"
export PT=14
for v in 3_3 6_3 6_6 3_6; do
  echo "== PT=14 NA_DB=$v"
  GROUP="128:128,128:64" timeout 60 ./convlab_pt14_$v 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch\|check"
  timeout 60 ./convlab_pt14_$v 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch\|check"
  timeout 60 ./convlab_pt14_$v 32 56 56 64 128 3 0 2000 v3 0 | grep "us/launch"
done
