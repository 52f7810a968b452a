#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r6_n
for i in $(seq 1 24); do timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -q -s -p no:cacheprovider -k "recorded_into_a_graph_draw" 2>&1 | grep -E "AssertionError|passed|failed" | head -3; done | sort | uniq -c | tee gpurun_out/r6_n/pytest24.txt
