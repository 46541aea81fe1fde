mkdir -p gpurun_out/r03final
timeout 2000 python -m pytest tests -m gpu -q -s --durations=25 > gpurun_out/r03final/gpu_tests.log 2>&1
tail -3 gpurun_out/r03final/gpu_tests.log
bash tools/collect_artifacts.sh r03final 2>&1 | tail -5
