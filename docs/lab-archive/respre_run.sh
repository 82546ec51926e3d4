cd popnet_amd/build
export NBUF=1
for r in 1 2; do for b in convlab_prev convlab; do
  printf "%-13s 112res " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 1 | grep "us/launch"
  printf "%-13s 112    " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 0 | grep "us/launch"
done; done
./convlab 32 112 112 64 64 3 1 20 v3 1 | grep check
cd ../..
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -1
for i in 1 2; do python3 bench.py --no-cpu-baseline --steps 400 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['achieved'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'], [ (k['kernel'][13:30], k['avg_launch_us']) for k in r['conv_stack']['by_kernel'][:2]])"; done
