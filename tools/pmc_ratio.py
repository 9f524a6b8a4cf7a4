"""HBM bytes per DP cell of every ksw kernel from the two rocprofv3 --pmc passes of tools/prof_bench.sh
(FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md prescribes) and the
per-class cell counts that bench.py printed in the same runs. Writes profiles/pmc_bytes_per_cell.json.
usage: pmc_ratio.py gpurun_out/prof_<tag>"""
import sqlite3, sys, os, json, glob, collections, re

src = sys.argv[1]


def counters(d, name):
    f = glob.glob(os.path.join(src, d, "*.db"))[0]
    db = sqlite3.connect(f)
    agg = collections.defaultdict(float)
    for k, v in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (name,)):
        agg[k] += v
    return agg


def classes(log):
    m = re.search(r'\{"metric".*\}', open(os.path.join(src, log)).read())
    cl = json.loads(m.group(0))["roofline"]["classes"]
    return cl


def key(kernel_name):        # "void ksw_dp_kernel<16, true, false>(...)" -> "ksw_dp_kernel<16,true,false>"
    m = re.search(r"(ksw_[a-z_]+kernel(<[^>]*>)?)", kernel_name)
    if not m:
        return None
    k = m.group(1)
    cm = re.match(r"ksw_chain_kernel<(\d+), (\w+), (\w+), (\w+)>", k)      # the EXACT variants of one chained-workgroup class share a class id (bench.py: "*")
    if cm:
        return "ksw_chain_kernel<%s, %s, %s, *>" % cm.groups()[:3]
    pm = re.match(r"ksw_pmulti_kernel<(\d+), (\d+)", k)      # the <CLIP, HASN> variants of one geometry are one class in bench.py
    return "ksw_pmulti_kernel<%s, %s>" % pm.groups() if pm else k              # (spelled like bench.py's class names, spaces included; stripe kernels keep their <BP, NWV, CLIP, HASN>)


# the --pmc passes serialise the dispatches: their durations are each kernel's time ALONE on the chip (no sharing with concurrent kernels)
try:
    db = sqlite3.connect(glob.glob(os.path.join(src, "pmc1", "*.db"))[0])
    rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
    tot = sum(r[2] for r in rows) or 1.0
    print("kernel durations with serialised dispatches (pmc1 pass): name, calls, total ms, avg us, share")
    for nm, calls, total, avg in rows[:24]:
        print("  %-60s %6d %10.1f %10.1f %5.1f %%" % (nm[:60], calls, total / 1e3, avg, 100.0 * total / tot))
except Exception as e:
    print("no dispatch table:", e)
fetch_raw, write_raw = counters("pmc1", "FETCH_SIZE"), counters("pmc2", "WRITE_SIZE")
c1, c2 = classes("pmc1.log"), classes("pmc2.log")
fetch, write = collections.defaultdict(float), collections.defaultdict(float)
for kn, v in fetch_raw.items():
    fetch[key(kn)] += v
for kn, v in write_raw.items():
    write[key(kn)] += v
out = {}
for k in set(list(fetch) + list(write)):
    kn = k
    if not k or k not in c1 or k not in c2 or not c1[k]["cells"]:
        continue
    rd = 2.0 * fetch.get(kn, 0.0) * 1024 / c1[k]["cells"]
    wr = write.get(kn, 0.0) * 1024 / c2[k]["cells"]
    out[k] = rd + wr
    print("%-36s read %.3f B/cell  write %.3f B/cell  (algorithmic: 1 B/cell written)" % (k, rd, wr))
m_ = re.search(r'"reads_per_step_per_gpu": (\d+)', open(os.path.join(src, "pmc1.log")).read())
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 0 --reads-per-step %s (%s), FETCH_SIZE x2 (gfx950), KiB units" % (m_.group(1) if m_ else "?", os.path.basename(src.rstrip("/"))),
           "bytes_per_cell": out}, open("profiles/pmc_bytes_per_cell.json", "w"), indent=1)

# ---- instruction issue per DP cell (third pass): SQ_INSTS_* count wave-instructions, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY quad-cycles ----
try:
    names = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")
    raw = {n: counters("pmc3", n) for n in names}
    c3 = classes("pmc3.log")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for n in names:
        for kn, v in raw[n].items():
            per[key(kn)][n] += v
    kern = {}
    for k, v in per.items():
        if not k or k not in c3 or not c3[k]["cells"]:
            continue
        cells = c3[k]["cells"]
        ins = v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"]
        kern[k] = {"valu_per_cell": v["SQ_INSTS_VALU"] / cells, "salu_per_cell": v["SQ_INSTS_SALU"] / cells, "valu_per_128_cells": 128 * v["SQ_INSTS_VALU"] / cells,
                   "salu_per_128_cells": 128 * v["SQ_INSTS_SALU"] / cells, "cycles_per_instruction_per_wave": 4.0 * v["SQ_WAVE_CYCLES"] / max(ins, 1.0),
                   "wave_cycles_waiting_frac": v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1.0), "wave_cycles_issue_stall_frac": v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1.0),
                   "wave_cycles_active_frac": v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1.0)}
        print("%-40s VALU %.1f SALU %.1f per 128 cells, %.1f cycles per instruction per wave, waiting %.0f %% stalled %.0f %% issuing %.0f %%" %
              (k, kern[k]["valu_per_128_cells"], kern[k]["salu_per_128_cells"], kern[k]["cycles_per_instruction_per_wave"], 100 * kern[k]["wave_cycles_waiting_frac"],
               100 * kern[k]["wave_cycles_issue_stall_frac"], 100 * kern[k]["wave_cycles_active_frac"]))
    json.dump({"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY pass of bench.py --steps 1 --warmup 0 "
                         "--reads-per-step %s (%s): wave-instructions per DP cell; cycle counters are quad-cycles (MI355X_MICROARCH.md)" % (m_.group(1) if m_ else "?", os.path.basename(src.rstrip("/"))),
               "kernels": kern}, open("profiles/insts_per_cell.json", "w"), indent=1)
except Exception as e:
    print("no instruction counter pass:", e)
