"""Prints the kernel sequence of the LAST decode step in a rocprofv3 kernel trace (the launches between the last two
greedy-sampling kernels): duration, grid, workgroup size, name.  usage: python tools/decode_step_sequence.py <kernel_trace.csv>"""
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'argmax' in n]
a,b=idx[-2],idx[-1]
tot=0
for r in rows[a+1:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=re.sub(r'\(.*','',r['Kernel_Name'])
    n=re.sub(r'^_ZN5mi355\d+','',n)[:70]
    print(f"{(e-s)/1e3:7.2f} us  grid {r['Grid_Size_X']:>7}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}  {n}")
    tot+=(e-s)/1e3
print('sum',tot, 'span',(int(rows[b]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/1e3, 'launches', b-a)
