"""MFMA busy per kernel from one rocprofv3 PMC pass of bench.py:
  cd /tmp; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv
      -d <dir> -- python3 bench.py --steps 1 --warmup 0 --cpu-forwards 0 --no-kernel-timers --sustained-seconds 0
  python tools/mfma_busy.py <dir>
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), summed over every launch of the kernel in the run (whole
launch incl. prologue / epilogue / idle CUs: the attention grid at one member is 216 workgroups on 256 CUs)."""
import csv, glob, os, re, sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
rows = []
for k, c in agg.items():
    if c.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
    ratio = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA", 0) > 0 else float("nan")
    rows.append((c["GRBM_GUI_ACTIVE"], k, n[k], busy, ratio))
for _, k, cnt, busy, ratio in sorted(rows, reverse=True)[:12]:
    print(f"{k[:52]:52s} launches {cnt:5d}  MFMA busy {100 * busy:5.1f} %   VALU instr / MFMA instr = {ratio:.2f}")
