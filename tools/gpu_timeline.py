"""GPU occupancy over time from a rocprofv3 --kernel-trace database: how much of the wall time has >= 1 kernel running, the
average number of concurrent kernels, and how often the running kernels together have fewer waves than the chip can hold.
usage: gpu_timeline.py <results.db> [t_skip_fraction]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
t0 = rows[0][1]; t1 = max(r[2] for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
lo = t0 + (t1 - t0) * skip
ev = []
for n, s, e, g, wg in rows:
    if e <= lo: continue
    s = max(s, lo)
    waves = g // 64                        # grid_x is in work-items
    ev.append((s, 1, waves, n)); ev.append((e, -1, -waves, n))
ev.sort(key=lambda x: (x[0], x[1]))
CAP = 256 * 8                              # 256 CUs x ~8 waves each as "comfortably full"
busy = under = 0; conc_t = 0.0; cur = 0; curw = 0; last = lo
for t, d, w, n in ev:
    dt = t - last
    if cur > 0:
        busy += dt; conc_t += cur * dt
        if curw < CAP: under += dt
    cur += d; curw += w; last = t
span = t1 - lo
print("span %.2f s | >=1 kernel running %.1f %% | avg concurrent kernels (when busy) %.2f | busy but < %d waves resident-equivalent: %.1f %% of span" % (span / 1e9, 100 * busy / span, conc_t / max(busy, 1), CAP, 100 * under / span))
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for n, s, e, g, wg in rows:
    if e <= lo: continue
    k = n.split('(')[0][:44]; agg[k][0] += 1; agg[k][1] += (e - max(s, lo)) / 1e9; agg[k][2] += g // 64
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-46s calls=%6d sum=%7.2f s  avg %.1f ms  avg waves/launch %d" % (k, v[0], v[1], v[1] / v[0] * 1e3, v[2] // v[0]))
