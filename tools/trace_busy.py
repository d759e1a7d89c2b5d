"""Kernel busy time vs wall time of the launch bursts in a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):
   python tools/trace_busy.py <output dir>   -> one line per burst (bursts are separated by > 2 ms of idle GPU)."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take last N kernels forming the last forward: find big gaps
st=[int(r['Start_Timestamp']) for r in rows]; en=[int(r['End_Timestamp']) for r in rows]
# split into bursts by gaps > 2 ms
bursts=[[0]]
for i in range(1,len(rows)):
    if st[i]-en[i-1]>2_000_000: bursts.append([])
    bursts[-1].append(i)
for b in bursts[-4:]:
    if not b: continue
    busy=sum(en[i]-st[i] for i in b); wall=en[b[-1]]-st[b[0]]
    print(len(b),'kernels wall %.3f ms busy %.3f ms'%(wall/1e6,busy/1e6))
