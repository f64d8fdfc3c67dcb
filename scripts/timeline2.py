"""Kernel timeline of the last few G-BRE steps from a rocprofv3 kernel-trace CSV: every kernel with its queue, start, end (us,
relative to the third-last evaluation kernel) -- what the build stream's kernels wait for."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [i for i, r in enumerate(rows) if 'evaluate_bre' in r['Kernel_Name']]
i0 = ev[-3]; t0 = int(rows[i0]['Start_Timestamp'])
i1 = ev[-1]
prev_end = {}
for r in rows[i0 - 2:i1 + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    q = r['Queue_Id']
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print("%9.1f %9.1f dur %7.1f gap %7.1f q=%s vgpr=%s grid=%s %s" % (s, e, e - s, gap, q, r['VGPR_Count'], r.get('Grid_Size', '?'), r['Kernel_Name'][:40]))
