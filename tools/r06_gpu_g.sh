#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_g; mkdir -p $O
(MPG_FWD_TWO_TERM=1 timeout -k 10 600 python3 tools/determinism.py) > $O/determinism_two_term.txt 2>&1 || { tail -5 $O/determinism_two_term.txt; exit 1; }
cat $O/determinism_two_term.txt
(MPG_FWD_TWO_TERM=1 timeout -k 10 400 python3 tools/soak.py) > $O/soak_two_term.txt 2>&1 || { tail -5 $O/soak_two_term.txt; exit 1; }
cat $O/soak_two_term.txt
