cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
EML_PAIR_PASS=1 timeout 900 python -m pytest tests/test_gpu_densenet.py -m gpu -q 2>&1 | tail -15 > gpurun_out/diag_pair_all.txt
timeout 300 python -m pytest tests/test_gpu_densenet.py -m gpu -q -k pair_pass 2>&1 | grep -E "worst|passed|failed|assert " > gpurun_out/diag_pair_one.txt
cat gpurun_out/diag_pair_all.txt gpurun_out/diag_pair_one.txt
