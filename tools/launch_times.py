"""Prints the duration of every kernel launch of a rocprofv3 --kernel-trace CSV, in start order.
usage: python tools/launch_times.py TRACE.csv [first] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[first:first + count]:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0) / 1e3:10.1f} us  {(b - a) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>9s}x{r.get('Grid_Size_Y', '?'):>4s}  {name}")
