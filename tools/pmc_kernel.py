"""Sum of every counter per kernel name from the rocprofv3 --pmc databases under a directory: python tools/pmc_kernel.py <dir> [name filter]"""
import sqlite3, sys, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    try:
        for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
            agg[k][c] += v
    except sqlite3.Error as e:
        print("skip", f, e)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(agg):
    if flt in k:
        print(k[:110])
        for c in sorted(agg[k]):
            print("    %-28s %.4e" % (c, agg[k][c]))
