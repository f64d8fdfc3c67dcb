"""One G-VPM step's kernels in order with gaps (rocprofv3 kernel-trace CSV)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [i for i, r in enumerate(rows) if 'gather_vpm_kernel' in r['Kernel_Name']]
i0, i1 = ev[-3], ev[-2]
t0 = int(rows[i0]['End_Timestamp'])
prev = t0
print("from the end of one gather kernel to the end of the next: %.1f us" % ((int(rows[i1]['End_Timestamp']) - t0) / 1e3))
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("gap %6.1f  run %7.1f  q=%s %s" % ((s - prev) / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Kernel_Name'][:50]))
    prev = max(prev, e)
