"""Timeline of one verifier batch from a TODHIP_DEBUG log (tools/chained_ticks.py with CHAINED_TICKS_RAW): when each tick started
and ended, and the host time between them. usage: tick_timeline.py <raw log>"""
import re, sys
t_prev = None
for line in open(sys.argv[1]):
    mc = re.match(r"\[todhip (\d+)\] consumed", line)
    if mc:
        print("%8.0f  consumed (+%.0f us after the tick's end)" % (float(mc.group(1)), float(mc.group(1)) - t_prev))
        t_prev = float(mc.group(1))
        continue
    line = re.sub(r" \((lane \d+), (\d+) slots\)", r" \1 \2", line)
    m = re.match(r"\[todhip (\d+)\] (tick ([0-9.]+) us: (.*)|flight of (\d+) slot\(s\)( landed)?(: (.*))?)", line)
    if not m:
        continue
    t = float(m.group(1))
    if m.group(3):
        dur = float(m.group(3)); start = t - dur
        what = " ".join(w for w, v in zip(m.group(4).split()[0::2], m.group(4).split()[1::2]) if v not in ("0", "0+0"))
        cnts = " ".join(v for v in m.group(4).split()[1::2] if v not in ("0", "0+0"))
        print("%8.0f  +%5.0f host | tick %6.0f us  %s [%s]" % (start, start - (t_prev if t_prev else start), dur, what, cnts))
        t_prev = t
    else:
        print("%8.0f  %s" % (t, m.group(2)[:110]))
