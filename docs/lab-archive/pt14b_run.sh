cd popnet_amd/build
export NBUF=1
printf "PT7  2x2 112res "; timeout 60 ./convlab 32 112 112 64 64 3 1 2000 v3 1 | grep "us/launch"
printf "PT7  2x2 112    "; timeout 60 ./convlab 32 112 112 64 64 3 1 2000 v3 0 | grep "us/launch"
printf "PT14 2x1 112res "; PT=14 timeout 60 ./convlab 32 112 112 64 64 3 2 2000 v3 1 | grep "us/launch\|check"
printf "PT14 2x1 112    "; PT=14 timeout 60 ./convlab 32 112 112 64 64 3 2 2000 v3 0 | grep "us/launch"
printf "PT7  level      "; GROUP="128:128,128:64" timeout 60 ./convlab 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch"
printf "PT14 level      "; PT=14 GROUP="128:128,128:64" timeout 60 ./convlab 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch"
printf "PT7  56         "; timeout 60 ./convlab 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch"
printf "PT14 56         "; PT=14 timeout 60 ./convlab 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch"
cd ../..
one() { python3 bench.py --no-cpu-baseline --steps 400 $* 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('PT14=%s' % os.environ.get('POPNET_CONV3_PT14'), d['value'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])" $*; }
one
POPNET_CONV3_PT14=1 one
POPNET_CONV3_PT14=2 one
