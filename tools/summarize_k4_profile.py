"""Digest gpurun_out/prof_k4_* (tools/profile_k4.sh) into profiles/<tag>_<engine>_kernel_stats.csv and <tag>_<engine>_pmc.json
(engine = k4x for the matrix-core kernel, k4 for the vector-ALU one; taken from the profiled bench line)."""
import csv, glob, json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.join(root, "gpurun_out")
log = open(os.path.join(out, "prof_k4_trace.log")).read()
bench = json.loads(re.findall(r"^\{.*\}$", log, re.M)[-1])
mfma = bench["roofline"]["kernel"] == "hamming_topk_mfma"
name = "k4x" if mfma else "k4"
# gpurun merges a call's outputs into gpurun_out/ without removing older ones: take the NEWEST file of every directory
newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]
stats = newest(glob.glob(os.path.join(out, "prof_k4_trace", "**", "*kernel_stats.csv"), recursive=True))
assert stats, "no kernel_stats.csv"
shutil.copy(stats[0], os.path.join(root, "profiles", "%s_%s_kernel_stats.csv" % (tag, name)))
# the timed launches = the instantiation of the engine's kernel with the most calls (k = 2, radius 35; which query-block count
# the launcher picked is part of the name; bench.py also runs a few launches of the other engine afterwards)
KERNEL, avg_ns, calls = None, None, 0
for row in csv.DictReader(open(stats[0])):
    if ("hamming_topk_mfma<" if mfma else "hamming_topk_tiles<") in row["Name"] and int(row["Calls"]) > calls:
        KERNEL = re.search(r"(hamming_topk_\w+<[^>]*>)", row["Name"]).group(1)
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
assert avg_ns, "matcher kernel not in the trace"

pmc = {}
for d in sorted(glob.glob(os.path.join(out, "prof_k4_pmc*"))):
    for f in newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc, n = {}, {}
        for row in csv.DictReader(open(f)):
            if KERNEL not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            n[c] = n.get(c, set()); n[c].add(row["Dispatch_Id"])
        for c in acc:
            pmc[c] = acc[c] / len(n[c])
# FETCH_SIZE / WRITE_SIZE are in KB. The guide's gfx950 correction: FETCH_SIZE reports half of the bytes of wide (16 B per lane)
# coalesced streaming reads -- the matrix-core kernel reads the DB exactly that way (global_load_dwordx4 per lane), the
# vector-ALU kernel reads it with 64-byte scalar loads (no correction). WRITE_SIZE is exact.
fetch = pmc.get("FETCH_SIZE", 0.0) * 1024.0 * (2.0 if mfma else 1.0)
traffic = fetch + pmc.get("WRITE_SIZE", 0.0) * 1024.0
queries = int(bench["roofline"]["queries_per_launch"])
rows = int(bench["config"]["db_rows_per_gpu"])
doc = {
    "kernel": KERNEL,
    "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --stages match "
               "--extras= (tools/profile_k4.sh; each counter set in its own pass; digested by tools/summarize_k4_profile.py)",
    "workload": "C3: %d queries (a step's batch of frames) x %d DB rows per launch, k=2, radius 35 on one MI355X" % (queries, rows),
    "queries_per_launch": queries,
    "avg_duration_ns_kernel_trace": avg_ns, "calls": calls, "pmc_per_launch": pmc,
    "hbm_traffic_bytes_per_launch": traffic,
    "fetch_correction": "x2 (16 B per lane coalesced loads, MI355X_MICROARCH.md HBM section)" if mfma else "none (64-byte scalar loads)",
    "clock_ghz": pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / avg_ns if avg_ns else None,   # summed over the 8 XCDs
    "bench_launch_ms_live": bench["roofline"]["launch_ms"],
}
if mfma:
    # every MFMA of this kernel is a v_mfma_f32_32x32x64_f8f6f4 = 32 matrix-pipe cycles; 1024 SIMDs
    busy = pmc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    cycles = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    doc["mfma_pipe_busy_fraction"] = busy / (1024.0 * cycles) if cycles else None
    doc["mfma_per_launch"] = pmc.get("SQ_INSTS_MFMA")
    doc["valu_insts_per_mfma"] = (pmc.get("SQ_INSTS_VALU", 0.0) - pmc.get("SQ_INSTS_MFMA", 0.0)) / max(pmc.get("SQ_INSTS_MFMA", 1.0), 1.0)
    peak = os.path.join(out, "mfma_fp4_peak.txt")
    if os.path.exists(peak):
        best = 0.0
        for line in open(peak):
            m = re.search(r"([0-9.]+) PFLOP/s, ([0-9.]+) T 256-bit pairs/s", line)
            if m:
                best = max(best, float(m.group(1)))
        shutil.copy(peak, os.path.join(root, "profiles", "%s_mfma_fp4_peak_microbench.txt" % tag))
        doc["measured_mfma_roof"] = {"PFLOPs": best, "source": "profiles/%s_mfma_fp4_peak_microbench.txt (tools/mfma_fp4_peak.hip: bare "
                                     "v_mfma_f32_32x32x64_f8f6f4 loop, operands in registers, random +-1 data, same box and run)" % tag,
                                     "kernel_fraction_of_it": queries * rows * 512.0 / (avg_ns * 1e-9) / 1e15 / best if best else None}
else:
    nq_waves = (queries + 63) // 64
    doc["valu_insts_per_row_and_wave"] = pmc.get("SQ_INSTS_VALU", 0.0) / (rows * nq_waves)
json.dump(doc, open(os.path.join(root, "profiles", "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k not in ("pmc_per_launch", "command")}, indent=1))
