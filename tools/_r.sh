timeout 1500 python -m pytest tests/test_bm_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/t_bm.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/t_bm.txt | tail -5
