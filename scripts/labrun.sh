cd pop-net_amd/build
export NBUF=2
for na in 3 6; do echo "== NA $na"; for s in "32 28 28 128 128 3 0" "32 28 28 256 256 3 0" "32 28 28 192 256 3 0" "32 56 56 128 128 3 0"; do ./convlab_na$na $s 50 v3 0 | tail -2; done; done
./convlab_stamp 32 28 28 256 256 3 0 5 v3 0 | grep stamps
