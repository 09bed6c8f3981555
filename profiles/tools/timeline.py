import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'felics' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
nl=int(sys.argv[2]) if len(sys.argv)>2 else 2
# the first kernel of a sub-batch: k_front's first slice (four slices per queued submission), k_wide_count for 16-bit samples
idx=[i for i,r in enumerate(rows) if 'k_front' in r['Kernel_Name']]
per=4
if not idx:
    idx=[i for i,r in enumerate(rows) if 'k_wide_count' in r['Kernel_Name']]
    per=1
start=idx[-nl*per]
last=rows[start:]
base=int(last[0]['Start_Timestamp'])
for r in last:
    name=r['Kernel_Name'].split('felics::')[1].split('<')[0].split('(')[0]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if d<0.02 and name not in ('k_spine',): continue
    print(f"{name:16s} q={r['Queue_Id']} start={(int(r['Start_Timestamp'])-base)/1e6:8.3f} end={(int(r['End_Timestamp'])-base)/1e6:8.3f} dur={d:7.3f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
