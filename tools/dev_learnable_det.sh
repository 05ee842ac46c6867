#!/bin/bash
# Developer tool (GPU box): is the learnable-task trajectory REPRODUCIBLE in deterministic mode?  N runs, one shared tuning cache,
# ZSG_DETERMINISTIC=1: every run's loss curve must equal the first one's bit for bit; prints the first differing step otherwise.
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-10}; STEPS=${2:-150}; T=/tmp/det_learn; mkdir -p $T; rm -f $T/*
export ZSG_DETERMINISTIC=1 ZSG_TUNE_CACHE=$T/tune.json
cd $R
python tools/dev_learnable.py --steps 3 --noeval > /dev/null 2>&1      # fills the tuning cache
for i in $(seq $N); do python tools/dev_learnable.py --steps $STEPS --noeval --save $T/c$i.npy 2>&1 | grep "^end"; done
python - <<P
import numpy as np, glob
a = np.load("$T/c1.npy")
for i in range(2, $N + 1):
    b = np.load("$T/c%d.npy" % i)
    d = np.nonzero(a != b)[0]
    print("run", i, "identical" if len(d) == 0 else f"first differs at step {d[0]}: {a[d[0]]!r} vs {b[d[0]]!r}; {len(d)} steps differ")
P
