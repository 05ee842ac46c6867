"""Developer tool (GPU box): greedy in-step refinement of a tuning cache.  Fresh tunings of one build disagree on a handful of launch shapes
(near-equal tiles by single-launch latency); which alternative is better INSIDE the two-stream step only the step can tell.  For every shape
on which the given caches disagree, every alternative choice is tried on top of the current best cache and kept when the step gets faster by
more than the noise (two confirming runs).
usage: python tools/refine_tuning.py <best.json> <other1.json> [...]   -> <best>.refined.json + a log on stdout
       ... --objective-forward      rank by bench.py's forward-only leg instead of the step (the step may not get slower); wgrad entries are skipped
       python tools/refine_tuning.py <best.json> --toggle-w8        ... every direct-kernel entry with the 8-wave workgroup bit flipped
       python tools/refine_tuning.py <best.json> --toggle24        the alternatives are every Winograd entry with tile_hint bit 24 flipped (forward /
                                                                   data gradient: four position groups <-> two; weight gradient: the block order)"""
import json
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD = "--objective-forward" in sys.argv     # rank by the forward-only leg (north_star's own quantity); the step must not get slower
BENCH = [sys.executable, os.path.join(R, "bench.py"), "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-roofline", "--other-configs", "off"] + (["--forward-leg"] if FWD else [])
TMP = os.path.join(R, "gpurun_out", "tunings", "_try.json")
NOISE = float(os.environ.get("REFINE_NOISE_MS", "0.015"))


def step_ms(cache: dict) -> float:
    json.dump(cache, open(TMP, "w"))
    env = dict(os.environ, ZSG_SHIPPED_TUNE="0", ZSG_TUNE_CACHE=TMP)
    out = subprocess.run(BENCH, env=env, capture_output=True, text=True).stdout
    m = re.search(r'"ms_per_step": ([0-9.]+)', out)
    if FWD and m:
        f = re.search(r'"forward": \{"median_ms": ([0-9.]+)', out)
        STEP[0] = float(m.group(1))
        return float(f.group(1)) if f else 1e9
    return float(m.group(1)) if m else 1e9


STEP = [0.0]


best = json.load(open(sys.argv[1]))
others = [json.load(open(p)) for p in sys.argv[2:] if not p.startswith("--")]
if "--toggle-w8" in sys.argv:      # direct kernels: the 8-wave workgroup variants (implicit GEMM: every tile; weight gradient: the 128x128 tile)
    others.append({k: v ^ (1 << 24) for k, v in best.items()
                   if not (v & 0x40000000) and ((k.startswith("('igemm'") and (v & 0xff) in (64, 128)) or (k.startswith("('wgrad'") and (v & 0xffff) == 0x8080))})
if "--toggle24" in sys.argv:
    others.append({k: v ^ (1 << 24) for k, v in best.items() if v & 0x40000000})
base = min(step_ms(best), step_ms(best))
base_step = STEP[0]
print(f"base {base:.3f} ms" + (f" (forward; step {base_step:.3f})" if FWD else ""), flush=True)
for k in sorted(best):
    if FWD and k.startswith("('wgrad'"):
        continue
    alts = sorted({o[k] for o in others if k in o and o[k] != best[k]})
    for v in alts:
        trial = dict(best)
        trial[k] = v
        t = step_ms(trial)
        verdict = ""
        if t < base - NOISE and (not FWD or STEP[0] < base_step + NOISE):
            t2 = step_ms(trial)
            if max(t, t2) < base - NOISE / 2 and (not FWD or STEP[0] < base_step + NOISE):
                best, base, verdict = trial, (t + t2) / 2, "  -> kept"
            else:
                verdict = f"  (second run {t2:.3f}: not kept)"
        print(f"{k[:96]}  -> {hex(v)}: {t:.3f} ms" + (f" (step {STEP[0]:.3f})" if FWD else "") + verdict, flush=True)
print(f"refined: {base:.3f} ms")
json.dump(best, open(sys.argv[1].replace(".json", ".refined.json"), "w"))
