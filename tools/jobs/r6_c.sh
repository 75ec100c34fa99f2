#!/bin/bash
# round 6, job c: operand-feed probe (request shapes of the LDS-DMA pieces vs bytes per clock per CU)
R="${GRAFT_REPO_ROOT:-.}"; O=$R/gpurun_out/r6_c; mkdir -p $O
cd $R
timeout 300 ./tools/probes/feed_probe > $O/r6_feed_probe.txt 2>&1
cat $O/r6_feed_probe.txt
