"""L2 hit rate of the batch-1 GEMM launches with and without the next-weight prefetch, from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass over
tools/forward_ab.py --batch 1 --variants 0 8388608 --rounds 1 (first the prefetch on, then off; the kernels are told apart by dispatch order).
Usage: python tools/pf_hitrate.py <counter_collection.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_disp = collections.OrderedDict()
for r in rows:
    d = int(r["Dispatch_Id"])
    e = by_disp.setdefault(d, {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = [v for _, v in sorted(by_disp.items())]
gemms = [v for v in disp if "gemm_l_kernel" in v["name"] or "gemm_lp_kernel" in v["name"] or "gemm_g_kernel" in v["name"]]
print(len(disp), "dispatches,", len(gemms), "GEMM launches")
half = len(gemms) // 2
for label, part in (("first half of the run (prefetch ON)", gemms[:half]), ("second half (prefetch OFF)", gemms[half:])):
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for v in part:
        m = re.search(r"gemm_\w+<(\d+)", v["name"])
        k = f"epilogue {m.group(1)}" if m else v["name"][:40]
        agg[k][0] += v.get("TCC_HIT_sum", 0.0)
        agg[k][1] += v.get("TCC_MISS_sum", 0.0)
        agg[k][2] += 1
    print(label)
    for k, (h, mi, n) in sorted(agg.items()):
        print(f"  {k:14s} launches {n:5d}  hits {h / n:12.0f}  misses {mi / n:12.0f} per launch  hit rate {h / max(h + mi, 1):.3f}")
