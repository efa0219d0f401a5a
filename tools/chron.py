import csv, re, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'assign_kernel' in r['Kernel_Name']]
spans=[(marks[i],marks[i+1]) for i in range(len(marks)-1)]
spans=[s for s in spans if 100<s[1]-s[0]<700][-10:]
spans.sort(key=lambda s:int(rows[s[1]]['Start_Timestamp'])-int(rows[s[0]]['Start_Timestamp']))
lo,hi=spans[len(spans)//2]
t0=int(rows[lo]['Start_Timestamp'])
def short(n):
    n=re.sub(r'^void\s+','',n).replace('(anonymous namespace)::','')
    return n[:60]
prev_end={}
for r in rows[lo:hi]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    q=r['Queue_Id']
    grid=int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))
    print("%8.1f %7.1f q%s g%-6d %s"%((s-t0)/1e3,(e-s)/1e3,q,grid*int(r.get('Grid_Size_Y',1) or 1)*int(r.get('Grid_Size_Z',1) or 1) if False else grid,short(r['Kernel_Name'])))
