"""Developer tool: overlap analysis of a rocprofv3 --kernel-trace CSV (one training step of the steady state).
usage: python tools/trace_overlap.py <kernel_trace.csv> [step_index_from_end] [--list]
--list also prints every launch of the step in start order: queue, start offset, duration, idle time of its queue before it.
Prints, for the chosen step: wall, GPU-busy union, time with >= 2 kernels in flight, time per kernel class while it runs ALONE."""
import csv
import re
import sys
from collections import defaultdict


def cls(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "")
    n = re.sub(r"<.*", "", n)
    if n.startswith(("igemm", "wino_kernel", "wgrad_kernel", "wino_wgrad")):
        return "mfma:" + n
    if n.startswith("bn_"):
        return "bn"
    return "other"


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    # steps are delimited by the adam kernel
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
    lo, hi = adam[-k - 1] + 1, adam[-k] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print(f"step: {len(step)} launches, wall {(t1 - t0) / 1e3:.1f} us, queues {sorted(set(r[3] for r in step))}")
    if "--list" in sys.argv:
        qend, qs = {}, sorted(set(r[3] for r in step))
        for s, e, n, q in step:
            gap = (s - qend[q]) / 1e3 if q in qend else 0.0
            qend[q] = e
            short = re.sub(r"\(.*", "", n).replace("void ", "")[:70]
            print(f"  q{qs.index(q)} +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short}")
    split = next(r[0] for r in step if "loss" in r[2])
    for name, a, b in (("forward", t0, split), ("backward+update", split, t1)):
        analyse(name, [r for r in step if a <= r[0] < b], a, b)


def analyse(name, step, t0, t1):
    print(f"--- {name}: {(t1 - t0) / 1e3:.1f} us, {len(step)} launches")
    ev = []
    for i, (s, e, n, q) in enumerate(step):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    live = set()
    last = t0
    busy = multi = 0
    alone = defaultdict(int)
    alone_k = defaultdict(int)
    pair = defaultdict(int)
    for t, d, i in ev:
        dt = t - last
        if live:
            busy += dt
            if len(live) >= 2:
                multi += dt
                key = "+".join(sorted(set(cls(step[j][2]).split(":")[0] for j in live)))
                pair[key] += dt
            else:
                alone[cls(step[next(iter(live))][2])] += dt
                alone_k[re.sub(r"\(.*", "", step[next(iter(live))][2])[:60]] += dt
        last = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    print(f"busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us, >=2 kernels in flight {multi / 1e3:.1f} us")
    print("alone:", {k_: round(v / 1e3, 1) for k_, v in sorted(alone.items(), key=lambda x: -x[1])})
    print("alone, non-MFMA kernels:", {k_: round(v / 1e3, 1) for k_, v in sorted(alone_k.items(), key=lambda x: -x[1]) if not k_.replace("void ", "").startswith(("igemm", "wino", "wgrad_k"))})
    print("co-running classes:", {k_: round(v / 1e3, 1) for k_, v in sorted(pair.items(), key=lambda x: -x[1])})
    tot = defaultdict(int)
    for s, e, n, q in step:
        tot[cls(n)] += e - s
    print("sum of durations:", {k_: round(v / 1e3, 1) for k_, v in sorted(tot.items(), key=lambda x: -x[1])})


if __name__ == "__main__":
    main()
