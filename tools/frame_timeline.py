"""Prints the launches of the last frame in a rocprofv3 kernel trace (CSV): start offset, duration and the
gap before each, to see where a frame's time goes between the kernels.
usage: python tools/frame_timeline.py KERNEL_TRACE_CSV"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
begins = [i for i, r in enumerate(rows) if "k_begin" in r["Kernel_Name"]]
rows = rows[begins[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    print(f"{(a - t0) / 1e3:9.1f} us  +{(a - prev_end) / 1e3:6.1f} gap  {(b - a) / 1e3:8.1f} us  {name}")
    prev_end = max(prev_end, b)
print(f"frame: {(prev_end - t0) / 1e3:.1f} us")
