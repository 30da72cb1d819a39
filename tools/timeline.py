"""Reads a rocprofv3 kernel-trace CSV and prints, for the last frames of the run, when each frame's stages
ran (ms from the first k_begin of the window): rounds, drain, sum; and how much of the window no kernel ran.
usage: python tools/timeline.py KERNEL_TRACE.csv [frames]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
begins = [i for i, r in enumerate(rows) if r[2] == "k_begin"]
first = begins[-frames]
t0 = rows[first][0]
win = rows[first:]
# busy time: union of kernel intervals
busy, cur_s, cur_e = 0, None, None
for s, e, n, q in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for s, e, n, q in win) - t0
print(f"window {span / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
for s, e, n, q in win:
    if n in ("k_begin", "k_finish", "k_reduce_head", "k_reduce_tail", "k_reduce", "k_mark_unfinished"):
        print(f"  {n:18s} queue {q}  {(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms")
# everything from the second-to-last drain's start to one millisecond later
marks = [s for s, e, n, q in win if n == "k_mark_unfinished"]
if len(marks) >= 2:
    print("  around a drain:")
    for s, e, n, q in win:
        if marks[-2] - 300000 <= s <= marks[-2] + 1200000:
            print(f"    {n[:40]:40s} queue {q}  {(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms")
ts = [(s, e) for s, e, n, q in win if n == "k_trace_shade"]
if ts:
    d = [(e - s) / 1e3 for s, e in ts]
    print(f"  k_trace_shade: {len(ts)} launches, mean {sum(d) / len(d):.1f} us, slowest {max(d):.1f} us, first five {[round(x) for x in d[:5]]}")
