cd pop-net_amd/build
export NBUF=1
for b in convlab_regst; do
echo $b
GROUP="128:128,128:64" timeout 20 ./$b 32 28 28 256 256 3 0 50 v3 0 | grep "us/launch\|check"
timeout 20 ./$b 32 28 28 256 256 3 0 50 v3 0 | grep "us/launch"
timeout 20 ./$b 32 28 28 192 128 3 0 50 v3 0 | grep "us/launch\|check"
timeout 20 ./$b 32 112 112 64 64 3 1 50 v3 1 | grep "us/launch\|check"
timeout 20 ./$b 32 56 56 128 128 3 0 50 v3 1 | grep "us/launch\|check"
timeout 20 ./$b 32 56 56 64 64 3 2 50 v3 1 | grep "us/launch\|check"
done
