# timing-only ablations of the split-bf16 forward tile kernel (wrong results by construction): where its time goes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for flags in "" "-DTX_FAKE_NOHALO" "-DTX_FAKE_ONEPASS" "-DTX_FAKE_NOHALO -DTX_FAKE_ONEPASS"; do
  POPNET_EXTRA_HIPCC_FLAGS="$flags" python3 popnet_amd/build.py --force > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x3abl -o t -- python3 scripts/train_bench.py 32 3 bf16x3 > /dev/null 2>&1
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/x3abl/t_kernel_stats.csv')):
    if 'tconv3_tile_x3' in r['Name'] or 'wgrad_x3' in r['Name']: print('flags [$flags]', r['Name'][:28], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))"
done
python3 popnet_amd/build.py --force > /dev/null 2>&1
