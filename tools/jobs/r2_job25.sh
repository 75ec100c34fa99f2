#!/bin/bash
# block-paged KV cache: kernel tests, e2e tests, then the whole attention/e2e files for regressions
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 -k "paged" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q --timeout 300 -k "paged" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_server.py -x -q --timeout 300 -k "attention or rope or e2e or generate or prefix or packed or server or batch" 2>&1 | tail -5
