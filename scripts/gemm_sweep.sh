#!/bin/bash
# usage: bash scripts/gemm_sweep.sh "0 1 2 ..." [lds_pad]
for v in $1; do ROHM_GEMM_LDS_PAD=${2:-0} ROHM_GEMM_VARIANT=$v python scripts/gemm_bench.py 2>&1 | grep variant; done
