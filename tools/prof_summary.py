"""Summarise rocprofv3 sqlite outputs (gpurun_out/prof_<tag>/{stats,pmc*}/ksw_results.db) into a text file
suitable for profiles/: per-kernel calls / total / average duration, and per-kernel sums of the PMC counters."""
import sqlite3, sys, os, collections, glob

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_" + tag
out = []
st = os.path.join(src, "stats")
for f in glob.glob(os.path.join(st, "*.db")):
    db = sqlite3.connect(f)
    out.append("== kernel-trace --stats (%s) ==" % f)
    out.append("%-110s %8s %14s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append("%-110s %8d %14.0f %12.0f %6.2f" % (r[0][:110], r[1], r[2], r[3], r[4]))
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    for f in glob.glob(os.path.join(d, "*.db")):
        db = sqlite3.connect(f)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        ndisp = collections.defaultdict(set)
        for k, c, v, disp in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            agg[k][c] += v
            ndisp[k].add(disp)
        out.append("")
        out.append("== --pmc pass (%s): counter sums over all dispatches ==" % f)
        for k in agg:
            out.append("%s  [dispatches=%d]" % (k[:120], len(ndisp[k])))
            out.append("    " + "  ".join("%s=%.6g" % (c, v) for c, v in sorted(agg[k].items())))
open(os.path.join("profiles", tag + ".txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
