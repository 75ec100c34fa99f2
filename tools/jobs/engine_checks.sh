#!/bin/bash
# the scheduler-facing GPU tests only (engine, server, paged KV, prefix reuse): a quick check after host-side changes
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_server.py -x -q --timeout 300 -k "engine or paged or prefix or packed or server or chat or batch or inference" 2>&1 | grep -v Warning | tail -12
