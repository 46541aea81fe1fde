"""Print the kernel timeline of the last forward found in a rocprofv3 kernel trace csv (development aid)."""
import csv, glob, os, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "chan_to_token" in r["Kernel_Name"]]
seq = rows[idx[int(os.environ.get("TRACE_START", "-2"))]:]  # TRACE_START=-1: the DCAE decode of tools/dcae_one.py
t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0
agg = {}
for r in seq:
    n = r["Kernel_Name"]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:30]
    key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg.setdefault(key, []).append((e - s) / 1e3)
    if len(sys.argv) > 2:
        print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f}  {short}  grid {key[1]}")
    prev_end = e
    if "token_to_chan" in n:
        break
busy = sum(sum(v) for v in agg.values())
print(f"forward total {(prev_end - t0) / 1e3:.1f} us, kernels {busy:.1f} us, gaps {(prev_end - t0) / 1e3 - busy:.1f} us over {sum(len(v) for v in agg.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k[0]:30s} grid {k[1]:6d}  x{len(v):2d}  avg {sum(v) / len(v):7.1f} us  total {sum(v):7.1f}")
