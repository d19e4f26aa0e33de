"""Digest gpurun_out/prof_k4_* (tools/profile_k4.sh) into profiles/<tag>_k4_kernel_stats.csv and <tag>_k4_pmc.json."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.join(root, "gpurun_out")
KERNEL = "hamming_topk_tiles<2, 2>"   # k = 2, three-stage schedule: the timed launches (bench.py also runs 3 dense ones afterwards)

stats = glob.glob(os.path.join(out, "prof_k4_trace", "**", "*kernel_stats.csv"), recursive=True)
assert stats, "no kernel_stats.csv"
shutil.copy(stats[0], os.path.join(root, "profiles", "%s_k4_kernel_stats.csv" % tag))
avg_ns = calls = None
for row in csv.DictReader(open(stats[0])):
    if KERNEL in row["Name"]:
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])

pmc = {}
for d in sorted(glob.glob(os.path.join(out, "prof_k4_pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, n = {}, {}
        for row in csv.DictReader(open(f)):
            if KERNEL not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            n[c] = n.get(c, set()); n[c].add(row["Dispatch_Id"])
        for c in acc:
            pmc[c] = acc[c] / len(n[c])
traffic = (pmc.get("FETCH_SIZE", 0.0) + pmc.get("WRITE_SIZE", 0.0)) * 1024.0
# the profiled command prints bench.py's JSON line: take the launch shape from it
import re
log = open(os.path.join(out, "prof_k4_trace.log")).read()
bench = json.loads(re.findall(r"^\{.*\}$", log, re.M)[-1])
queries = int(bench["roofline"]["queries_per_launch"])
rows = int(bench["config"]["db_rows_per_gpu"])
nq_waves = (queries + 63) // 64
doc = {
    "kernel": "hamming_topk_tiles<2>",
    "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --stages match "
               "(tools/profile_k4.sh; each counter set in its own pass; digested by tools/summarize_k4_profile.py)",
    "workload": "C3: %d queries (a step's batch of frames) x %d DB rows per launch, k=2, radius 35 on one MI355X" % (queries, rows),
    "queries_per_launch": queries,
    "avg_duration_ns_kernel_trace": avg_ns, "calls": calls, "pmc_per_launch": pmc,
    "hbm_traffic_bytes_per_launch": traffic,
    "valu_insts_per_row_and_wave": pmc.get("SQ_INSTS_VALU", 0.0) / (rows * nq_waves),
    "clock_ghz": pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / avg_ns if avg_ns else None,   # summed over the 8 XCDs
    "note": "FETCH_SIZE/WRITE_SIZE are in KB. The DB rows are read with 64-byte scalar loads (s_load_dwordx16), not with "
            "wide coalesced vector loads, so the guide's x2 correction for 16 B/lane streams does not apply: FETCH_SIZE "
            "matches the 32.0 MB of DB rows + 32 KB of queries read once per launch; WRITE_SIZE is the per-tile partial "
            "lists. SQ_INSTS_VALU / (DB rows x 64-query waves) = VALU instructions per row and wave: 8 xor + 8 popcount "
            "for a full 256-bit distance, 4 + 4 when the 128-bit lower bound already rules the row out. "
            "GRBM_GUI_ACTIVE / 8 XCDs / duration = clock.",
}
json.dump(doc, open(os.path.join(root, "profiles", "%s_k4_pmc.json" % tag), "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("avg_duration_ns_kernel_trace", "calls", "hbm_traffic_bytes_per_launch", "valu_insts_per_row_and_wave", "clock_ghz")}))
