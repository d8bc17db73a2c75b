"""Build container: gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (start/end timestamps in ns).
   python3 tools/trace_gaps.py <dir with *_kernel_trace.csv> [max rows]"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
prev_end = None
out = []
for s, e, k in rows:
    out.append((k, (e - s) / 1e3, None if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
# print the last n rows (the timed regions are at the end of a bench run)
for k, d, g in out[-n:]:
    print("%-42s dur %7.2f us   gap before %s" % (k, d, "   -" if g is None else "%8.2f us" % g))
