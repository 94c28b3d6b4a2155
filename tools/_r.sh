timeout 1500 python -m pytest tests/test_sgm_gpu.py -m gpu -x -q > gpurun_out/t_sgm.txt 2>&1; grep -E "passed|failed|error" gpurun_out/t_sgm.txt | tail -3
