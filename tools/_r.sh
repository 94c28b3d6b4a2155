SGM_PATH_MODE=0 TAG=r06_default timeout 250 bash tools/pmc_sgm_dirs.sh > /dev/null
SGM_PATH_MODE=33 TAG=r06_round5 timeout 250 bash tools/pmc_sgm_dirs.sh > /dev/null
SGM_PATH_MODE=0 timeout 100 python tools/time_sgm.py 2048 2054 128 > gpurun_out/time_sgm_r06.txt 2>&1
SGM_PATH_MODE=33 timeout 100 python tools/time_sgm.py 2048 2054 128 > gpurun_out/time_sgm_r06_round5.txt 2>&1
cat gpurun_out/time_sgm_r06.txt
