#!/bin/bash
# Developer tool (GPU box): N complete fresh tunings of the headline configuration, each kept as a tuning cache and re-timed twice in fresh
# processes; the cache with the fastest step seeds the shipped table (tools/make_tuning_table.py --seed).  The tuner ranks the candidates of
# ONE launch by latency (median of interleaved samples); fresh tunings of one build still differ by ~0.5 % of the step, because near-equal tiles
# behave differently beside the other stream's kernels — that part is only visible in the step itself.
#   tools/best_of_tunings.sh [N=6] [tag] -> gpurun_out/tunings<tag>/{c<i>.json, summary.txt, best.json}     (AB_ARGS: another configuration, e.g.
#   AB_ARGS="--backbone ssd_vgg --bs 32" tools/best_of_tunings.sh 4 _ssd)
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-6}; O=$R/gpurun_out/tunings$2; mkdir -p $O; cd $R
B="python bench.py --steps ${AB_STEPS:-100} --warmup 20 --no-cpu-baseline --no-roofline --other-configs off ${AB_ARGS:-}"
G='"ms_per_step": [0-9.]*'
: > $O/summary.txt
for i in $(seq $N); do
  rm -f $O/c$i.json
  ZSG_SHIPPED_TUNE=0 ZSG_TUNE_CACHE=$O/c$i.json $B > /dev/null 2>&1
done
for rep in 1 2; do for i in $(seq $N); do
  echo "c$i $(ZSG_SHIPPED_TUNE=0 ZSG_TUNE_CACHE=$O/c$i.json $B 2>/dev/null | grep -o "$G" | head -1)" >> $O/summary.txt
done; done
python - <<PY
import collections
t = collections.defaultdict(list)
for l in open("$O/summary.txt"):
    p = l.split()
    if len(p) >= 3: t[p[0]].append(float(p[2]))
m = {k: sum(v) / len(v) for k, v in t.items()}
best = min(m, key=m.get)
open("$O/summary.txt", "a").write("# mean ms/step: " + " ".join(f"{k}={v:.3f}" for k, v in sorted(m.items())) + f"\n# best: {best}\n")
import shutil; shutil.copy("$O/" + best + ".json", "$O/best.json")
PY
python3 zsgnet-pytorch_amd/csrc/stamp.py | grep -o '"[0-9a-f]*"' | tr -d '"' > $O/best.stamp      # -> tools/_tunings/seed.stamp beside seed.json
cat $O/summary.txt
