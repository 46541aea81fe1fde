#!/bin/bash
# Measurement artefacts of one build, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   gpurun -- 'bash tools/collect_artifacts.sh r03f'          everything
#   gpurun -- 'bash tools/collect_artifacts.sh r03f pmc'      only the two fabric-byte PMC passes + pmc_summary.json (after a kernel edit)
# rocprofv3: the program itself follows `--` (no env / bash -c hop); counters in their own passes with --kernel-trace only.
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
Q="--cpu-forwards 0 --sustained-seconds 0"
pmc_passes() {
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_FETCH_SIZE -- python3 $B --steps 1 --warmup 0 $Q --no-kernel-timers > $O/pmc_fetch_run.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_WRITE_SIZE -- python3 $B --steps 1 --warmup 0 $Q --no-kernel-timers > $O/pmc_write_run.log 2>&1
  (cd $R && python3 tools/summarize_pmc.py gpurun_out $O/pmc_summary.json > $O/pmc_summary.txt 2>&1)
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
}
if [ "${2:-all}" = pmc ]; then
  pmc_passes
  cat $O/pmc_summary.txt
  exit 0
fi
python3 $B --steps 20 --warmup 5 > $O/bench_cfg2_bf16x3.json 2> $O/bench.err
python3 $B --no-batched-conditioning $Q > $O/bench_cfg2_bf16x3_conditioning_per_evaluation.json 2>> $O/bench.err
python3 $B --precision bf16 $Q > $O/bench_cfg2_bf16.json 2>> $O/bench.err
python3 $B --members-per-gpu 2 --lead-steps 40 --steps 3 --warmup 1 $Q --no-kernel-timers > $O/bench_cfg3_share_2members_40leadsteps.json 2>> $O/bench.err
python3 $B --model 1.6B --lead-steps 10 --steps 3 --warmup 1 $Q --no-kernel-timers > $O/bench_cfg4_share_1p6B_10leadsteps.json 2>> $O/bench.err
python3 $B --decode --lead-steps 40 --precision bf16 --steps 3 --warmup 1 $Q --no-kernel-timers > $O/bench_cfg5_share_decode_40leadsteps_bf16.json 2>> $O/bench.err
python3 $B --decode --lead-steps 40 --steps 3 --warmup 1 $Q --no-kernel-timers > $O/bench_cfg5_share_decode_40leadsteps_bf16x3.json 2>> $O/bench.err
python3 $B --workload dcae --cpu-forwards 0 > $O/dcae_encode_decode.json 2>> $O/bench.err
python3 $B --ensemble-size 16 --lead-steps 40 --steps 2 --warmup 0 $Q --no-kernel-timers > $O/bench_cfg3_whole_job_16members_one_gpu.json 2>> $O/bench.err
# the headline leg ALONE under rocprofv3: per template instance, AverageNs x bench.py's flops_per_launch = achieved_rocprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --steps 20 --warmup 5 $Q --no-kernel-timers > $O/stats_run.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $B --steps 2 --warmup 1 $Q --no-kernel-timers > $O/trace_run.log 2>&1
python3 $R/tools/trace_forward.py $O/trace v > $O/forward_timeline.txt 2>&1
pmc_passes
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/mfma -- python3 $B --steps 1 --warmup 0 $Q --no-kernel-timers > $O/mfma_run.log 2>&1
python3 $R/tools/mfma_busy.py $O/mfma > $O/mfma_busy.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/dtrace -- python3 $R/tools/dcae_one.py 1 > $O/dcae_trace_run.log 2>&1
TRACE_START=-1 python3 $R/tools/trace_forward.py $O/dtrace v > $O/dcae_decode_1frame_timeline.txt 2>&1
rm -rf $O/dtrace
# keep the merged scratch small: the raw traces are large
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_cfg2_headline.csv \;
rm -rf $O/stats $O/trace $O/mfma
cd $R
LDC_LIB_PATH=ladcast_amd/libladcast_hip_stamps.so python3 tools/gemm_launch_stamps.py > $O/gemm_launch_stamps.log 2>&1
ls -la $O
