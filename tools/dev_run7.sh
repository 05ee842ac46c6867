#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
python tools/w4_kscan.py > $O/w4_kscan.txt 2>&1
ZSG_W4_ABL=3 python tools/w4_kscan.py >> $O/w4_kscan.txt 2>&1
