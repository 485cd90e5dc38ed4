"""Timeline of the last kernels of a rocprofv3 --kernel-trace run (csv): start, duration and the gap to the previous kernel's end.
usage: python tools/trace_gaps.py <dir> [n_last]"""
import csv, glob, sys
root = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# the steady part: skip the cpu_baseline etc.; take a window in the middle of the render launches
idx = [i for i, r in enumerate(rows) if "s2r_render_kernel" in r[2]]
mid = idx[len(idx) // 2]
w = rows[mid - 2: mid - 2 + n_last]
t0 = w[0][0]
prev_end = None
for s, e, name, q in w:
    print("%9.1f us  dur %7.1f us  gap %6.1f us  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev_end) / 1e3) if prev_end else 0.0, q, name))
    prev_end = e if prev_end is None else max(prev_end, e)
