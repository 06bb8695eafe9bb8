import csv,re,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'spconv' in r['Kernel_Name'] or 'conv1' in r['Kernel_Name']]
n=len(sel)//max(1,sum('conv1' in r['Kernel_Name'] for r in sel))     # launches per forward (one first convolution each)
tot=0
for r in sel[-n:]:
    nm=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).replace('void ','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(f"{re.sub(r'[(].*','',nm)[:50]:50s} grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']:>4s} lds {r['LDS_Block_Size']:>7s} {d:9.1f} us")
print("total",tot)
