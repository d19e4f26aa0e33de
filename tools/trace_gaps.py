"""What runs between two DB passes of the matcher: gaps on the matcher's queue from a rocprofv3 --kernel-trace CSV (tools/trace_dist1.sh)."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ham = [r for r in rows if "hamming_topk_mfma" in r["Kernel_Name"]][-22:-2]
q = ham[0]["Queue_Id"]
gaps, own = [], {}
for a, b in zip(ham[:-1], ham[1:]):
    gaps.append((b["s"] - a["e"]) / 1e3)
    for r in rows:
        if r["Queue_Id"] == q and r["s"] >= a["e"] and r["e"] <= b["s"]:
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:32]
            own.setdefault(n, []).append((r["e"] - r["s"]) / 1e3)
print("DB pass %.0f us (median), gap to the next %.0f us (median, %d gaps); kernels of the same queue inside the gaps:" % (
    statistics.median((r["e"] - r["s"]) / 1e3 for r in ham), statistics.median(gaps), len(gaps)))
for n, v in sorted(own.items(), key=lambda kv: -sum(kv[1])):
    print("   %-34s %4.1f per gap, %6.1f us each" % (n, len(v) / len(gaps), statistics.median(v)))
