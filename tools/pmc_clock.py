"""per-kernel MFMA-busy rate and candidate clock figures from one rocprofv3 PMC pass with --kernel-trace (development aid):
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py ..."""
import csv, glob, sys
from collections import defaultdict

root = sys.argv[1]
cc = glob.glob(root + "/*/*counter_collection.csv")[0]
kt = glob.glob(root + "/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); tot = defaultdict(float)
for r in csv.DictReader(open(cc)):
    d = dur.get(r["Dispatch_Id"])
    if not d:
        continue
    name = d[1].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] + " g%d" % d[2]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[name] += 1; tot[name] += d[0]
for k in sorted(tot, key=lambda k: -tot[k])[:10]:
    a, t = acc[k], tot[k]
    wave_res = a["SQ_WAVE_CYCLES"] / max(a["SQ_WAVES"], 1)  # resident cycles per wave (counter unit: cycles, some parts count in 4s)
    print("%-46s n=%4d avg %7.1f us | GRBM/8/t %5.0f MHz | MFMA busy/1024/t %5.0f MHz | wave residency %8.0f cyc = %5.0f MHz x avg t | waves/launch %.0f"
          % (k, n[k], t / n[k] / 1e3, a["GRBM_GUI_ACTIVE"] / 8 / t * 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / t * 1e3, wave_res, wave_res / (t / n[k]) * 1e3,
             a["SQ_WAVES"] / n[k]))
