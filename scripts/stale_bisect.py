"""Does any back-end result depend on device memory this handle has not written, or on what an earlier window left in a slot?  (GPU)

1. runs scripts/stale_probe.py with device buffers that start as zeros (the product's behaviour) and checks the identities the probe is built around
   (a window behind other windows in the same handle == the window in a fresh handle);
2. runs it again with every fresh device buffer of gf_ba filled with PLAUSIBLE stale data (GF_BA_POISON=4: what hipMalloc hands back behind another handle);
3. for every scenario whose digest moved: bisects over the allocations of the process (GF_BA_POISON_RANGE), names the buffer (GF_BA_ALLOC_TRACE) and narrows the
   dependence down to the first / last element of that buffer whose content matters (GF_BA_POISON_ELEMS).
python scripts/stale_bisect.py [--budget SECONDS] [--out FILE]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "scripts", "stale_probe.py")
IDENTITIES = [("plain_after", "plain"), ("gnss_after", "gnss"), ("slots", "slots_fresh")]
T0 = time.time()
LOG = []


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    LOG.append(line)


def run(names, **env):
    e = dict(os.environ, GF_NO_TORCH_PRELOAD="1")
    for k in ("GF_BA_POISON", "GF_BA_POISON_RANGE", "GF_BA_POISON_ELEMS", "GF_BA_ALLOC_TRACE"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, PROBE] + list(names), capture_output=True, text=True, cwd=ROOT, env=e)
    res = {}
    for ln in out.stdout.splitlines():
        parts = ln.split()
        if len(parts) == 2:
            res[parts[0]] = parts[1]
    for nm in names:
        res.setdefault(nm, "ERR:" + out.stderr.strip()[-300:].replace("\n", " | "))
    return res, out.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=1200.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--scenarios", default="plain,plain_after,gnss,gnss_after,slots,slots_fresh,group,group_gnss")
    a = ap.parse_args()
    names = a.scenarios.split(",")
    left = lambda: a.budget - (time.time() - T0)

    ref, _ = run(names)
    say("== zero-initialised buffers (the product)")
    for nm in names:
        say("  %-12s %s" % (nm, ref[nm]))
    ref2, _ = run(names)
    say("== run-to-run:", "identical" if ref2 == ref else "DIFFERENT " + str({k: (ref[k], ref2[k]) for k in names if ref[k] != ref2[k]}))
    for x, y in IDENTITIES:
        if x in ref and y in ref:
            say("== identity %-12s == %-12s : %s" % (x, y, "holds" if ref[x] == ref[y] else "VIOLATED"))
    pois, _ = run(names, GF_BA_POISON=4)
    moved = [nm for nm in names if pois[nm] != ref[nm]]
    say("== plausible stale data in every fresh device buffer (GF_BA_POISON=4): moved:", moved if moved else "nothing")
    for nm in moved:
        say("  %-12s %s" % (nm, pois[nm]))
    garb, _ = run(names, GF_BA_POISON=2)
    say("== 0x5A garbage in every fresh device buffer (GF_BA_POISON=2): moved:", [nm for nm in names if garb[nm] != ref[nm]] or "nothing")
    nan, _ = run(names, GF_BA_POISON=5)
    moved5 = [nm for nm in names if nan[nm] != ref[nm]]
    say("== 0xFF (NaN / -1) in every fresh device buffer (GF_BA_POISON=5): moved:", moved5 or "nothing")
    PM = 4
    if moved5:     # 0xFF is the sharper probe (every double a NaN): bisect with it where it shows anything
        moved, PM = moved5, 5

    for nm in moved:
        if left() < 60:
            say("budget spent before", nm)
            break
        single, _ = run([nm])      # the bisection runs one scenario per process: its reference is the same scenario alone
        ref[nm] = single[nm]
        chk, _ = run([nm], GF_BA_POISON=PM)
        if chk[nm] == ref[nm]:
            say("== %s does not move when it runs alone (mode %d): skipped" % (nm, PM))
            continue
        _, err = run([nm], GF_BA_POISON=PM, GF_BA_POISON_RANGE="100000:100001", GF_BA_ALLOC_TRACE=1)
        allocs = [ln for ln in err.splitlines() if ln.startswith("gf_ba alloc")]
        n = len(allocs)
        say("== bisecting %s over %d allocations" % (nm, n))
        bad = []

        def rec(lo, hi):
            if left() < 30 or len(bad) >= 6:
                return
            r, _ = run([nm], GF_BA_POISON=PM, GF_BA_POISON_RANGE="%d:%d" % (lo, hi))
            if r[nm] == ref[nm]:
                return
            if hi - lo == 1:
                bad.append(lo)
                return
            mid = (lo + hi) // 2
            rec(lo, mid)
            rec(mid, hi)
        rec(0, n)
        for i in bad:
            say("  allocation %d matters: %s" % (i, allocs[i] if i < n else "?"))
            try:
                count = int(allocs[i].split("  ")[1].split(" x ")[0])
            except Exception:
                continue

            def differs(lo, hi):
                r, _ = run([nm], GF_BA_POISON=PM, GF_BA_POISON_RANGE="%d:%d" % (i, i + 1), GF_BA_POISON_ELEMS="%d:%d" % (lo, hi))
                return r[nm] != ref[nm]
            for prefer_left in (True, False):
                lo, hi = 0, count
                while hi - lo > 1 and left() > 30:
                    mid = (lo + hi) // 2
                    first, second = ((lo, mid), (mid, hi)) if prefer_left else ((mid, hi), (lo, mid))
                    if differs(*first):
                        lo, hi = first
                    elif differs(*second):
                        lo, hi = second
                    else:
                        say("    (the dependence needs elements of both halves of [%d, %d))" % (lo, hi))
                        break
                say("    %s element range that matters: [%d, %d) of %d" % ("first" if prefer_left else "last", lo, hi, count))
    say("== done in %.0f s" % (time.time() - T0))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write("\n".join(LOG) + "\n")


if __name__ == "__main__":
    main()
