import csv,sys,collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:40]))
rows.sort()
# take the last 60% of the trace (timed region)
n=len(rows); rows=rows[int(n*0.5):]
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for i,(s,e,k) in enumerate(rows):
    dur[k].append(e-s)
    if i: gap[k].append(s-rows[i-1][1])
tot=rows[-1][1]-rows[0][0]
print("span us",tot/1e3,"kernels",len(rows))
for k in dur:
    d=dur[k]; g=gap[k] or [0]
    print(f"{k:42s} n={len(d):6d} avg={sum(d)/len(d)/1e3:8.2f}us  gap_before={sum(g)/len(g)/1e3:7.2f}us  share={(sum(d)+sum(g))/tot*100:5.1f}%")
