cd popnet_amd/build
for nb in 1 4; do export NBUF=$nb
  printf "NBUF=$nb 28x28 256->128 "; timeout 60 ./convlab 32 28 28 256 128 1 0 3000 v3 0 | grep "us/launch\|check" | tr '\n' ' '; echo
  printf "NBUF=$nb 56x56 128->128 "; timeout 60 ./convlab 32 56 56 128 128 1 0 3000 v3 0 | grep "us/launch\|check" | tr '\n' ' '; echo
  printf "NBUF=$nb 56x56  64->128 "; timeout 60 ./convlab 32 56 56 64 128 1 0 3000 v3 0 | grep "us/launch\|check" | tr '\n' ' '; echo
done
NBUF=1 timeout 60 ./convlab_stamp 32 28 28 256 128 1 0 200 v3 0 | grep "stamps"
NBUF=4 timeout 60 ./convlab_stamp 32 28 28 256 128 1 0 200 v3 0 | grep "stamps"
