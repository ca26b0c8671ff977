"""GPU idle-gap report from a rocprofv3 --kernel-trace CSV: busy time, idle time and the (previous kernel -> next kernel)
pairs that own the idle time, over the last `window_ms` of the trace.
usage: gap_report.py <dir with *kernel_trace.csv> [window_ms] [min_gap_us]"""
import collections, csv, glob, sys
f = [x for x in glob.glob(sys.argv[1] + "/**/*.csv", recursive=True) if "kernel_trace" in x][0]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
end = ev[-1][1]
sel = [e for e in ev if e[0] >= end - win * 1e6]
span = sel[-1][1] - sel[0][0]
busy = 0
cur_end = sel[0][0]
gaps, gapn = collections.Counter(), collections.Counter()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
prev = None
for s, e, n in sel:
    if s > cur_end:
        g = s - cur_end
        if prev is not None and g >= ming * 1e3:
            k = short(prev) + " -> " + short(n)
            gaps[k] += g
            gapn[k] += 1
    busy += max(0, e - max(s, cur_end))
    if e > cur_end:
        cur_end, prev = e, n
print(f"window {span/1e6:.1f} ms: busy {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms, {len(sel)} kernels")
for k, v in gaps.most_common(30):
    print(f"{v/1e6:8.2f} ms {gapn[k]:6d} x {v/gapn[k]/1e3:7.1f} us  {k}")
