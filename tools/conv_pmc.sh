#!/bin/bash
# Fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and MFMA busy of ONE conv shape, per library variant.
#   gpurun -- 'bash tools/conv_pmc.sh <tag> "<cin cout H W frames>" <lib suffix ...>'   ("" = the shipped library)
set -u
TAG=$1; SHAPE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for SUF in "$@"; do
  if [ "$SUF" = "shipped" ]; then unset LDC_LIB_PATH; else export LDC_LIB_PATH=$R/ladcast_amd/libladcast_hip_$SUF.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_$C -- python3 $R/tools/conv_one.py $SHAPE > $O/run_${SUF}_$C.log 2>&1
  done
  rm -rf $O/p_M
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/p_M -- python3 $R/tools/conv_one.py $SHAPE > $O/run_${SUF}_M.log 2>&1
  python3 - "$O" "$SUF" "$SHAPE" <<'PY' | tee -a $O/conv_pmc_summary.txt
import csv, glob, sys
O, suf, shape = sys.argv[1:4]
def mean(cdir, counter, keys=("conv_halo_kernel", "gemm_bf16x3_v3_kernel")):
    v = []
    for f in glob.glob(f"{O}/{cdir}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in keys):
                v.append(float(r["Counter_Value"]))
    return sum(v) / max(len(v), 1), len(v)
f, n = mean("p_FETCH_SIZE", "FETCH_SIZE"); w, _ = mean("p_WRITE_SIZE", "WRITE_SIZE")
mb, _ = mean("p_M", "SQ_VALU_MFMA_BUSY_CYCLES"); gui, _ = mean("p_M", "GRBM_GUI_ACTIVE"); iv, _ = mean("p_M", "SQ_INSTS_VALU"); im, _ = mean("p_M", "SQ_INSTS_MFMA")
alg = open(f"{O}/run_{suf}_FETCH_SIZE.log").read().strip().splitlines()[-1]
print(f"[{suf}] conv {shape}: {n} launches, FETCH_SIZE raw {f/1024:.1f} MiB (x2 = {2*f*1024/1e6:.1f} MB) + WRITE_SIZE {w*1024/1e6:.1f} MB -> fabric bytes per launch {(2*f+w)*1024/1e6:.1f} MB; "
      f"MFMA busy {100*mb/(gui/8*4*256) if gui else 0:.1f} % of SIMD-cycles (busy {mb:.3g}, GUI_ACTIVE {gui:.3g}); VALU/MFMA instr {iv/max(im,1):.2f}\n    {alg}")
PY
  rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_M
done
