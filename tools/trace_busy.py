import csv,glob,sys,collections
k=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')
kr=sorted(csv.DictReader(open(k[0])), key=lambda r:int(r['Start_Timestamp']))
ours=[r for r in kr if any(x in r['Kernel_Name'] for x in ('k_glm','k_firth','k_job','k_repack','k_af','k_bitdot','k_readback','rocclr'))]
# steady state: the last 60 % of the kernels
ours=ours[int(len(ours)*0.4):]
t0=int(ours[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in ours)
busy=0; cs=int(ours[0]['Start_Timestamp']); ce=int(ours[0]['End_Timestamp'])
for r in ours[1:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
nb=sum(1 for r in ours if 'k_repack_bits' in r['Kernel_Name'])
print('%s: span %.1f ms, busy %.1f ms (%.0f %%), batches %d -> %.2f ms per batch' % (sys.argv[1], (t1-t0)/1e6, busy/1e6, 100*busy/(t1-t0), nb, (t1-t0)/1e6/max(nb,1)))
tot=collections.Counter()
for r in ours: tot[r['Kernel_Name'].replace('void ','').split('(')[0][:30]]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
print('   kernel time per batch (ms):', [(k,round(v/1e6/nb,2)) for k,v in tot.most_common(7)])
