"""Between two evaluations on the gather stream: what runs, how long, how long nothing runs (rocprofv3 kernel-trace CSV)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [r for r in rows if 'evaluate_bre' in r['Kernel_Name']]
q = ev[-1]['Queue_Id']
same = [r for r in rows if r['Queue_Id'] == q]
gaps = []
for a, b in zip(ev[-9:-1], ev[-8:]):
    ta, tb = int(a['End_Timestamp']), int(b['Start_Timestamp'])
    mid = [r for r in same if ta <= int(r['Start_Timestamp']) < tb]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in mid)
    gaps.append((tb - ta, busy, [(r['Kernel_Name'][:28], (int(r['Start_Timestamp']) - ta) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in mid]))
for g in gaps:
    print("gap %.1f us, kernels in it %.1f us:" % (g[0] / 1e3, g[1] / 1e3), g[2])
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ev[-8:]]
print("evaluation %.1f us avg; period %.1f us" % (sum(d) / len(d) / 1e3, (int(ev[-1]['Start_Timestamp']) - int(ev[-9]['Start_Timestamp'])) / 8e3))
