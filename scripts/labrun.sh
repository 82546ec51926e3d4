cd pop-net_amd/build
export NBUF=1
for cfg in 0 4; do
GROUP="128:128,128:64" timeout 20 ./convlab 32 28 28 256 256 3 $cfg 50 v3 0 | grep "us/launch\|check"
timeout 20 ./convlab 32 28 28 256 256 3 $cfg 50 v3 0 | grep "us/launch"
timeout 20 ./convlab 32 56 56 128 128 3 $cfg 50 v3 1 | grep "us/launch"
done
