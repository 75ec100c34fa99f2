#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_server.py -x -q --timeout 300 -k "paged or prefix or packed or server or chat" 2>&1 | grep -v Warning | tail -15
