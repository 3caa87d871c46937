#!/usr/bin/env python3
"""tools/iter_timeline.py <prof_kernel_trace.csv>: the kernels of the last complete training iteration in a rocprofv3 trace
(start offset, duration, grid, name), and the sum of their durations."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_q_sample' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:6.1f}  grid {r['Grid_Size_X']:>7},{r['Grid_Size_Y']},{r['Grid_Size_Z']}  {name}")
print(f"{b - a} launches, {tot/1e3:.1f} us of kernel time")
