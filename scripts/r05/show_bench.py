"""Prints the figures of one bench.py JSON line that the round's targets are stated in."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
rf = d['roofline']
print('%s: %s value %.0f (%s) ms/step %.4f | frac %.3f (%.2f us) conv_stack %.3f %.3f ms' % (sys.argv[1].split('/')[-1], d['dtype'], d['value'], d.get('value_stat', {}).get('runs'), d['ms_per_step'], rf['frac'], rf['avg_launch_us'], rf['conv_stack']['frac'], rf['conv_stack']['ms_per_step']))
if 'h2d_inclusive' in d:
    h = d['h2d_inclusive']; print('   h2d %.0f = %.3f of value, link %s GB/s' % (h['value'], h['fraction_of_value'], (h.get('host_link') or {}).get('GBps')))
print('   postproc %s us/step' % d.get('postproc', {}).get('us_per_step'))
for k in rf['conv_stack']['by_kernel']:
    print('     %-44s x%-4.1f %8.2f us/step  %7.2f us  %7.1f TF/s' % (k['kernel'], k['launches_per_step'], k['us_per_step'], k['avg_launch_us'], k['tflops']))
pm = d.get('parity_mode') or {}
if 'value' in pm:
    print('   parity', pm['value'], pm['roofline']['frac'], (pm.get('h2d_inclusive') or {}).get('value'))
if 'value' in (d.get('yolo') or {}):
    print('   yolo', d['yolo']['value'], d['yolo']['conv_stack']['frac'])
if 'ms_per_step' in (d.get('train_step') or {}):
    print('   train', d['train_step']['ms_per_step'], d['train_step'].get('bf16x3', {}).get('ms_per_step'))
if 'rccl_check' in d:
    print('   rccl', {k: d['rccl_check'].get(k) for k in ('ranks_seen', 'gather_ms', 'gather_records_ok', 'error')})
print('   config keys:', sorted(k for k in d['config'] if k.startswith(('parity', 'h2d', 'value_', 'train', 'yolo'))))
print('   legs', d.get('leg_seconds'), 'same_records', d.get('frame_stats', {}).get('same_batch_same_records_across_steps_slots_and_input_modes'))
