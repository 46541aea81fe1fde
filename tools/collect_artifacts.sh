#!/bin/bash
# Measurement artefacts of one build, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   gpurun -- 'bash tools/collect_artifacts.sh r06x'          everything (~12 GPU-minutes)
#   gpurun -- 'bash tools/collect_artifacts.sh r06x pmc'      only the fabric-byte PMC passes + pmc_summary*.json (after a kernel edit)
# rocprofv3: the program itself follows `--` (no env / bash -c hop); counters in their own passes with --kernel-trace only.
# Round 6: the default line's `value` is the exact-fp32 mode; every artefact exists per arithmetic mode (fp32 = the headline, bf16x3 = the fast mode).
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
Q="--cpu-forwards 0 --sustained-seconds 0"
LEG="$Q --no-kernel-timers"   # the timed region alone
pmc_passes() {  # $1 = precision, $2 = output json
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_FETCH_SIZE -- python3 $B --precision $1 --steps 1 --warmup 0 $LEG > $O/pmc_fetch_run_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_WRITE_SIZE -- python3 $B --precision $1 --steps 1 --warmup 0 $LEG > $O/pmc_write_run_$1.log 2>&1
  (cd $R && python3 tools/summarize_pmc.py gpurun_out $O/$2 > $O/${2%.json}.txt 2>&1)
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
}
if [ "${2:-all}" = pmc ]; then
  pmc_passes fp32 pmc_summary_fp32.json
  pmc_passes bf16x3 pmc_summary.json
  cat $O/pmc_summary_fp32.txt $O/pmc_summary.txt
  exit 0
fi
# the default line exactly as the driver runs it (fp32 headline + bf16x3_mode + dcae + cfg5 + rccl_world1 + both cpu_baselines)
python3 $B --steps 20 --warmup 5 > $O/bench_default_line.json 2> $O/bench.err
python3 $B --precision bf16x3 --steps 20 --warmup 5 $Q --no-dcae-block --no-cfg5-block --no-rccl-world1 > $O/bench_cfg2_bf16x3.json 2>> $O/bench.err
python3 $B --precision bf16 $Q --no-dcae-block --no-cfg5-block --no-rccl-world1 > $O/bench_cfg2_bf16.json 2>> $O/bench.err
for P in fp32 bf16x3; do
  python3 $B --precision $P --members-per-gpu 2 --lead-steps 40 --steps 3 --warmup 1 $LEG > $O/bench_cfg3_share_2members_40leadsteps_$P.json 2>> $O/bench.err
  python3 $B --precision $P --model 1.6B --lead-steps 10 --steps 3 --warmup 1 $LEG > $O/bench_cfg4_share_1p6B_10leadsteps_$P.json 2>> $O/bench.err
  python3 $B --precision $P --ensemble-size 16 --lead-steps 40 --steps 2 --warmup 0 $LEG > $O/bench_cfg3_whole_job_16members_one_gpu_$P.json 2>> $O/bench.err
done
python3 $B --decode --lead-steps 40 --precision bf16 --steps 3 --warmup 1 $LEG > $O/bench_cfg5_share_decode_40leadsteps_bf16.json 2>> $O/bench.err
python3 $B --decode --lead-steps 40 --precision bf16x3 --steps 3 --warmup 1 $LEG > $O/bench_cfg5_share_decode_40leadsteps_bf16x3.json 2>> $O/bench.err
python3 $B --decode --lead-steps 40 --precision bf16x3 --decode-batch-frames 0 --steps 3 --warmup 1 $LEG > $O/bench_cfg5_share_decode_40leadsteps_bf16x3_per_chunk_decode.json 2>> $O/bench.err
python3 $B --workload dcae --cpu-forwards 0 > $O/dcae_encode_decode.json 2>> $O/bench.err
for P in fp32 bf16x3; do
  # the headline leg ALONE under rocprofv3: per template instance, AverageNs x bench.py's flops_per_launch = achieved_rocprof
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$P -- python3 $B --precision $P --steps 20 --warmup 5 $LEG > $O/stats_run_$P.log 2>&1
  find $O/stats_$P -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_cfg2_$P.csv \;
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$P -- python3 $B --precision $P --steps 2 --warmup 1 $LEG > $O/trace_run_$P.log 2>&1
  python3 $R/tools/trace_forward.py $O/trace_$P v > $O/forward_timeline_$P.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/mfma_$P -- python3 $B --precision $P --steps 1 --warmup 0 $LEG > $O/mfma_run_$P.log 2>&1
  python3 $R/tools/mfma_busy.py $O/mfma_$P > $O/mfma_busy_$P.txt 2>&1
  rm -rf $O/stats_$P $O/trace_$P $O/mfma_$P
done
pmc_passes fp32 pmc_summary_fp32.json
pmc_passes bf16x3 pmc_summary.json
rocprofv3 --kernel-trace --output-format csv -d $O/dtrace -- python3 $R/tools/dcae_one.py 1 > $O/dcae_trace_run.log 2>&1
TRACE_START=-1 python3 $R/tools/trace_forward.py $O/dtrace v > $O/dcae_decode_1frame_timeline.txt 2>&1
rm -rf $O/dtrace
cd $R
ls -la $O
