cd popnet_amd/build
export NBUF=1
for r in 1 2; do for b in convlab convlab_fastep; do
  printf "%-15s level  " $b; GROUP="128:128,128:64" timeout 60 ./$b 32 28 28 256 256 3 0 2000 v3 0 | grep "us/launch"
  printf "%-15s 112res " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 1 | grep "us/launch"
  printf "%-15s 112    " $b; timeout 60 ./$b 32 112 112 64 64 3 1 2000 v3 0 | grep "us/launch"
  printf "%-15s 56     " $b; timeout 60 ./$b 32 56 56 128 128 3 0 2000 v3 0 | grep "us/launch"
done; done
./convlab_fastep 32 112 112 64 64 3 1 20 v3 1 | grep check
./convlab_fastep 32 28 28 256 256 3 0 20 v3 0 | grep check
GROUP="128:128,128:64" timeout 60 ./convlab_stamp 32 28 28 256 256 3 0 200 v3 0 | grep "us/launch\|stamps"
timeout 60 ./convlab_stamp 32 112 112 64 64 3 1 200 v3 1 | grep "us/launch\|stamps"
cd ../..
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do python3 bench.py --no-cpu-baseline --steps 400 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['achieved'], r['frac'], r['conv_stack']['achieved'], r['conv_stack']['tflops_inside_timed_region'])"; done
python3 bench.py --no-cpu-baseline --steps 400 --net yolo 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('yolo', d['value'])"
