import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
g=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'conv3_kernel<1' in n or 'conv_mfma_kernel<1, 1' in n or 'Lb1' in n:
        key=(n[:60], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('LDS_Block_Size') or r.get('LDS_Block_Size_v'))
        g[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(g.items()):
    v=sorted(v); print(k, len(v), 'median %.1f us'%(v[len(v)//2]/1e3), 'min %.1f'%(v[0]/1e3))
