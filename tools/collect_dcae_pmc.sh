#!/bin/bash
# Counter evidence for the DCAE's conv / small-kernel launches: two fabric-byte PMC passes (FETCH_SIZE,
# WRITE_SIZE; each in its own run, --kernel-trace only), one MFMA-busy pass and a kernel-stats run of tools/dcae_one.py.
#   gpurun -- 'bash tools/collect_dcae_pmc.sh r04a 1'      (tag, frames)
set -u
TAG=${1:-run}
FR=${2:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="$R/tools/dcae_one.py"
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_FETCH_SIZE -- python3 $P $FR > $O/dcae_pmc_fetch_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_WRITE_SIZE -- python3 $P $FR > $O/dcae_pmc_write_run.log 2>&1
(cd $R && PMC_KEEP_TEMPLATE=1 python3 tools/summarize_pmc.py gpurun_out $O/dcae_pmc_summary_${FR}frame.json > $O/dcae_pmc_summary_${FR}frame.txt 2>&1)
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/mfma -- python3 $P $FR > $O/dcae_mfma_run.log 2>&1
python3 $R/tools/mfma_busy.py $O/mfma > $O/dcae_mfma_busy_${FR}frame.txt 2>&1
rm -rf $O/mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $P $FR > $O/dcae_stats_run.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/dcae_kernel_stats_${FR}frame.csv \;
rm -rf $O/stats
cat $O/dcae_pmc_summary_${FR}frame.txt $O/dcae_mfma_busy_${FR}frame.txt
