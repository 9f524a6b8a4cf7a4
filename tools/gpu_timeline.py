"""GPU occupancy over time from a rocprofv3 --kernel-trace database, restricted to the MAPPING PHASE (first ksw DP kernel .. last kernel; the
bench's set-up — reference / read generation, index build — has no kernels and would otherwise dilute every number): share of the wall time
with >= 1 kernel running, with and without the gaps between mini-batches, average number of concurrent kernels, the queued waves, and
the distribution of the concurrency. usage: gpu_timeline.py <results.db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
dp = [r for r in rows if r[0].startswith("void ksw_dpp") or "ksw_multi" in r[0]]
t0 = dp[0][1] if dp else rows[0][1]
t1 = max(r[2] for r in rows)
ev = []
for n, s, e, g in rows:
    if e <= t0:
        continue
    ev.append((max(s, t0), 1, g // 64)); ev.append((e, -1, -(g // 64)))      # grid_x is in work-items
ev.sort(key=lambda x: (x[0], x[1]))
busy = conc = wt = 0.0; cur = cw = 0; last = t0; gaps = []; hist = collections.Counter()
for t, d, w in ev:
    dt = t - last
    hist[min(cur, 16)] += dt
    if cur > 0:
        busy += dt; conc += cur * dt; wt += cw * dt
    elif dt > 2e8:
        gaps.append((last - t0, dt))
    cur += d; cw += w; last = t
span = t1 - t0
g = sum(x for _, x in gaps)
print("mapping phase %.2f s, %d kernels | >=1 kernel running %.1f %% of it, %.1f %% outside the %d gap(s) > 0.2 s between mini-batches (%.2f s) | avg concurrent kernels when busy %.2f | avg queued+resident waves when busy %.0f (the chip holds 8192..16384)"
      % (span / 1e9, len(rows), 100 * busy / span, 100 * busy / max(span - g, 1), len(gaps), g / 1e9, conc / max(busy, 1), wt / max(busy, 1)))
tot = sum(hist.values())
print("time share by number of concurrent kernels: " + ", ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for n, s, e, gx in rows:
    if e <= t0:
        continue
    k = n.split('(')[0][:44]; agg[k][0] += 1; agg[k][1] += (e - max(s, t0)) / 1e9; agg[k][2] += gx // 64
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print("  %-46s calls=%6d sum=%7.2f s  avg %.1f ms  avg waves/launch %d" % (k, v[0], v[1], v[1] / v[0] * 1e3, v[2] // v[0]))
