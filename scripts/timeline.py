"""Print the kernel timeline (>= 15 us kernels) of the last steps from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [i for i, r in enumerate(rows) if 'evaluate_bre' in r['Kernel_Name']]
i0 = ev[-4]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0 - 10:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    if e - s > 15 or 'plan' in r['Kernel_Name']:
        print("%9.1f %9.1f %7.1f q=%s lds=%s vgpr=%s %s" % (s, e, e - s, r['Queue_Id'], r['LDS_Block_Size'], r['VGPR_Count'], r['Kernel_Name'][:44]))
