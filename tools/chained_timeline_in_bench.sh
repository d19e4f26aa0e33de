# one verifier batch's timeline inside the running chained pipeline (TODHIP_DEBUG=1): usage chained_timeline_in_bench.sh [workers]
cd "$GRAFT_REPO_ROOT"
TODHIP_DEBUG=1 timeout -k 10 300 python bench.py --extras chained --stages match --no-cpu-baseline --verify-workers ${1:-2} --chained-workers ${1:-2} --steps 30 --repeats 1 > gpurun_out/tl.json 2> gpurun_out/tl.err
python3 - <<'PY'
import re, collections
lines = open("gpurun_out/tl.err").read().splitlines()
by = collections.defaultdict(list)
for ln in lines:
    m = re.search(r"\{t(\d+)\}$", ln)
    if m: by[int(m.group(1))].append(re.sub(r" \{t\d+\}$", "", ln))
for t, ls in sorted(by.items()):
    # the last complete batch of the thread: from the last 'lookup' tick on
    idx = [i for i, l in enumerate(ls) if re.search(r"tick .*lookup 1[0-9] ", l)]
    if len(idx) < 3: continue
    seg = ls[idx[-2]:idx[-1]]
    open("gpurun_out/tl_t%d.log" % t, "w").write("\n".join(seg) + "\n")
    print("thread", t, "lines", len(seg))
PY
for f in gpurun_out/tl_t*.log; do echo "== $f"; python tools/tick_timeline.py $f | grep -v consumed; done
