#!/bin/bash
# round 6, job g: release form decided at exchange creation (LLM(tensor_parallel_size=2) on one device), tp tests
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_tp_release.py -m gpu -x -q > $O/pytest_release.txt 2>&1
tail -30 $O/pytest_release.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_tp_p2p.py tests/test_gpu_kernels.py -m gpu -x -q -k "tiled or tp or bulk or exchange or allreduce" > $O/pytest_tp.txt 2>&1
tail -5 $O/pytest_tp.txt | cut -c1-300
