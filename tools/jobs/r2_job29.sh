#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_server.py tests/test_gpu_tp_p2p.py -x -q --timeout 300 2>&1 | tail -4
