#!/bin/bash
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_tp_p2p.py -q --timeout 300 2>&1 | tail -45
done
