"""Per-kernel summary (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev - the columns of rocprofv3's *_kernel_stats.csv) computed
from a rocprofv3 --kernel-trace CSV.  Fallback for runs in which `rocprofv3 --stats` itself crashes while writing its summary (round 6: the exact-fp32
headline leg, three attempts): the kernel trace of the SAME command is complete, and this is the same arithmetic over it.
usage: python tools/kernel_stats_from_trace.py <dir with *kernel_trace.csv> <out.csv>"""
import csv, glob, statistics, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
with open(sys.argv[2], "w", newline="") as out:
    w = csv.writer(out, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), f"{sum(v) / len(v):.6f}", f"{100.0 * sum(v) / tot:.6f}", min(v), max(v), f"{statistics.pstdev(v):.6f}"])
print(f"{len(d)} kernels, {sum(len(v) for v in d.values())} dispatches, {tot / 1e6:.1f} ms of kernel time from {f}")
