#!/usr/bin/env python3
"""Per-launch view of the decode-attention kernel in a rocprofv3 results DB (rocpd sqlite): launches in start order, duration against the
bytes the launch streams (its position in the decode sequence gives the context length), outliers, per-layer and per-step means.
usage: attn_decode_trace.py results.db B S [layers]"""
import sqlite3
import sys

db, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = int(sys.argv[4]) if len(sys.argv) > 4 else 32
c = sqlite3.connect(db)
rows = c.execute("select start, duration, name from kernels where name like '%attn_decode_kernel%' order by start").fetchall()
print(f"{len(rows)} launches of attn_decode_kernel")
# the probe calls generate() several times: split the launches into runs at start gaps > 20 ms, report the longest run
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[0] > 20e6:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
print("runs:", [len(r) for r in runs])
run = max(runs, key=len)
run = run[: len(run) // L * L]
H, d = 32, 128
tot = 0.0
fr = []
for i, (st, dur, _) in enumerate(run):
    step, layer = divmod(i, L)
    ctx = S + step + 1
    nbytes = 2.0 * B * ctx * H * d * 2
    fr.append((nbytes / dur, dur / 1e3, step, layer))        # bytes per ns = GB/s
    tot += dur
bw = sorted(f[0] for f in fr)
n = len(bw)
print(f"run of {n} launches: total {tot / 1e6:.2f} ms, mean {tot / n / 1e3:.1f} us; GB/s percentiles: min {bw[0]:.0f} p1 {bw[n // 100]:.0f} p10 {bw[n // 10]:.0f} "
      f"p50 {bw[n // 2]:.0f} p90 {bw[n * 9 // 10]:.0f} max {bw[-1]:.0f}")
slow = [f for f in fr if f[0] < 0.9 * bw[n // 2]]
print(f"{len(slow)} launches below 0.9 x median rate; time lost to them {sum(f[1] - f[1] * f[0] / bw[n // 2] for f in slow) / 1e3:.2f} ms")
for f in slow[:20]:
    print(f"   step {f[2]:3d} layer {f[3]:2d}: {f[1]:.1f} us, {f[0]:.0f} GB/s")
by_layer = [0.0] * L
for f in fr:
    by_layer[f[3]] += f[0]
steps = n // L
print("mean GB/s by layer:", " ".join(f"{x / steps:.0f}" for x in by_layer))
# gap between the end of the kernel before and the start of this kernel is not in this table (kernels only); start-to-start of consecutive attention launches:
gaps = [(run[i + 1][0] - run[i][0]) / 1e3 for i in range(n - 1)]
gs = sorted(gaps)
print(f"start-to-start of consecutive attention launches: p50 {gs[len(gs) // 2]:.1f} us, p90 {gs[len(gs) * 9 // 10]:.1f} us, max {gs[-1]:.1f} us")
by_step = {}
for f in fr:
    by_step.setdefault(f[2], []).append(f[0])
print("mean GB/s by step (every 8th):", " ".join(f"{k}:{sum(v) / len(v):.0f}" for k, v in sorted(by_step.items()) if k % 8 == 0))
