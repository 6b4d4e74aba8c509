#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_chain.py -q -x 2>&1 | tail -3 || exit 1
bash tools/r06_gpu_ab.sh oldgemm 3
