timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_h3.py -x -q -m gpu > gpurun_out/t_chain2.log 2>&1; tail -4 gpurun_out/t_chain2.log
ROUNDS=5 STEPS=10 python benchmarks/ab_step.py "wgrad_chain=0" "wgrad_chain=1" 2>&1 | grep median
CONFIG=causal ROUNDS=4 STEPS=10 python benchmarks/ab_step.py "wgrad_chain=0" "wgrad_chain=1" 2>&1 | grep median
