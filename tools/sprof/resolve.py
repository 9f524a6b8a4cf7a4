"""python tools/sprof/resolve.py sprof.txt [top_n]: self time per function from the samples of tools/sprof/libsprof.so, using `nm` on the modules that
exist on this machine (the repo's libraries travel to the GPU box unchanged, so their offsets resolve here)."""
import bisect, collections, os, subprocess, sys

def symtab(path):
    try:
        out = subprocess.run(["nm", "-C", "--defined-only", "-n", path], capture_output=True, text=True).stdout
        if not out.strip():
            out = subprocess.run(["nm", "-C", "-D", "--defined-only", "-n", path], capture_output=True, text=True).stdout
    except Exception:
        return [], []
    addrs, names = [], []
    for l in out.splitlines():
        p = l.split(" ", 2)
        if len(p) == 3 and p[1] in "tTwW":
            addrs.append(int(p[0], 16)); names.append(p[2])
    return addrs, names

def main():
    f = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    per_mod = collections.Counter(); per_fn = collections.Counter(); tabs = {}; total = 0
    hints = collections.defaultdict(list); main_exe = None
    for l in open(f):
        if l.startswith("#sym "):
            _, path, off, name = l.split()
            hints[os.path.basename(path)].append((int(off, 16), name + " [resolved]"))
        elif l.startswith("#main "):
            main_exe = l.split(" ", 1)[1].strip()
    stk = []
    for l in open(f):
        if l.startswith("#stk "):
            _, n, rest = l.rstrip("\n").split(" ", 2)
            stk.append((int(n), [x.rsplit(" ", 1) for x in rest.split(" | ")]))
    for l in open(f):
        if l.startswith("#"):
            continue
        path, off, n = l.rsplit(" ", 2)
        off = int(off, 16); n = int(n); total += n
        mod = os.path.basename(path)
        per_mod[mod] += n
        local = path if os.path.exists(path) else None
        if path == "[main]" and main_exe and os.path.exists(main_exe):
            local = main_exe
        if local is None:
            for cand in ("winnowmap_amd/" + mod, "oracle/_ref/" + mod):
                if os.path.exists(cand):
                    local = cand
        if local is None:
            per_fn[(mod, "?")] += n; continue
        if local not in tabs:
            a_, n_ = symtab(local)
            merged = sorted(list(zip(a_, n_)) + hints.get(mod, []))
            tabs[local] = ([x[0] for x in merged], [x[1] for x in merged])
        a, names = tabs[local]
        i = bisect.bisect_right(a, off) - 1
        per_fn[(mod, names[i] if i >= 0 else "?")] += n
    print("total samples %d = %.1f CPU-s" % (total, total / 1e3))
    print("-- by module")
    for m, n in per_mod.most_common(12):
        print("  %6.2f %%  %8.1f s  %s" % (100.0 * n / total, n / 1e3, m))
    print("-- by function (self time)")
    for (m, fn), n in per_fn.most_common(top):
        print("  %6.2f %%  %8.1f s  %-22s %s" % (100.0 * n / total, n / 1e3, m, fn[:150]))

    # callers: for the hottest leaf functions, the code addresses found on the stack above the sample (nearest symbols; not an unwound call chain)
    if stk:
        def sym(path, off):
            mod = os.path.basename(path)
            local = path if os.path.exists(path) else None
            if path == "[main]" and main_exe and os.path.exists(main_exe):
                local = main_exe
            if local is None:
                for cand in ("winnowmap_amd/" + mod, "oracle/_ref/" + mod):
                    if os.path.exists(cand):
                        local = cand
            if local is None:
                return mod + ":?"
            if local not in tabs:
                a_, n_ = symtab(local)
                merged = sorted(list(zip(a_, n_)) + hints.get(mod, []))
                tabs[local] = ([x[0] for x in merged], [x[1] for x in merged])
            a, names = tabs[local]
            i = bisect.bisect_right(a, int(off, 16)) - 1
            return mod.split(".")[0] + ":" + (names[i][:60] if i >= 0 else "?")
        leaf = collections.defaultdict(collections.Counter); leaf_tot = collections.Counter()
        for n, chain in stk:
            names = [sym(p_, o_) for p_, o_ in chain]
            leaf_tot[names[0]] += n
            leaf[names[0]][" <- ".join(names[1:5])] += n
        print("-- stack words above the hottest leaves (nearest symbols)")
        for lf, n in leaf_tot.most_common(8):
            print("  %s  (%d samples)" % (lf, n))
            for ch, m in leaf[lf].most_common(6):
                print("      %5d  %s" % (m, ch))

if __name__ == "__main__":
    main()
