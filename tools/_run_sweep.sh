mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_search_gpu.py -x -q -k "fp32_scores_do_not_depend" > gpurun_out/t_f32s.log 2>&1; echo "rc=$?" >> gpurun_out/t_f32s.log; tail -3 gpurun_out/t_f32s.log
timeout -k 10 900 python tools/quick_search_bench.py > gpurun_out/sweep_r03.log 2>&1; echo "rc=$?" >> gpurun_out/sweep_r03.log; cat gpurun_out/sweep_r03.log
