"""Developer tool: per-kernel LDS bank-conflict share from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES run.
usage: python tools/lds_conflict_survey.py <counter_collection.csv>"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        cnt[k] += 1
rows = sorted(((v.get("SQ_LDS_BANK_CONFLICT", 0.0), k, v) for k, v in agg.items()), reverse=True)
for c, k, v in rows[:16]:
    act = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{k:46s} launches {cnt[k]:5d}  conflict {c / 1e6:8.1f} M  lds_active {act / 1e6:8.1f} M  ratio {c / max(1.0, act):.3f}")
