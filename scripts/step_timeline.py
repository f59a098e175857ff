"""What the GPU does during the timed steps of bench.py: from a rocprofv3 kernel trace (start / end stamps of every kernel), the share of the wall clock with no kernel
running, with one of this library's families running alone, and with both families at once.

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/bench.py --steps 40 --warmup 5 --no-e2e --no-small-batch --no-pcie --no-cpu-baseline
  python scripts/step_timeline.py /tmp/tl [steps]

The timed region is taken as the span of the LAST `steps` launches of lk_track_kernel (one per step) up to the end of the last back-end kernel behind them -- before the
isolated passes bench.py appends (those run one family at a time and are cut off at the first gap of more than 20 ms)."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = []
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r["Kernel_Name"]
            fam = "ba" if ("gfb::" in n or "copy_list_kernel<0>" in n) else "tracker" if ("gf::" in n or "copy_list_kernel<1>" in n) else "other"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam, n.split("(")[0].replace("void ", "")))
    rows.sort()
    lk = [i for i, r in enumerate(rows) if "lk_track_kernel" in r[3]]
    # the main loop: consecutive LK launches less than 20 ms apart, the longest such run that contains warm-up + steps launches
    runs, cur = [], [lk[0]]
    for a, b in zip(lk, lk[1:]):
        if rows[b][0] - rows[a][0] < 20_000_000:
            cur.append(b)
        else:
            runs.append(cur); cur = [b]
    runs.append(cur)
    main_run = max(runs, key=len)
    first = main_run[-steps] if len(main_run) >= steps else main_run[0]
    t0 = rows[first][0]
    # the end: the last kernel before the first idle gap of more than 20 ms behind the last LK launch of the run
    t1 = rows[main_run[-1]][1]
    for r in rows[main_run[-1]:]:
        if r[0] - t1 > 20_000_000:
            break
        t1 = max(t1, r[1])
    ev = []
    for s, e, fam, _ in rows:
        if e <= t0 or s >= t1 or fam == "other":
            continue
        ev.append((max(s, t0), 1, fam)); ev.append((min(e, t1), -1, fam))
    ev.sort()
    cnt = {"ba": 0, "tracker": 0}
    acc = {"idle": 0, "ba only": 0, "tracker only": 0, "both": 0}
    gaps = []
    last = t0
    for t, dlt, fam in ev:
        state = "both" if cnt["ba"] and cnt["tracker"] else "ba only" if cnt["ba"] else "tracker only" if cnt["tracker"] else "idle"
        acc[state] += t - last
        if state == "idle" and t - last > 0:
            gaps.append(t - last)
        last = t
        cnt[fam] += dlt
    acc["idle"] += t1 - last
    n = min(steps, len(main_run))
    span = t1 - t0
    print("timed region by the trace: %d steps, %.3f ms per step" % (n, span / n / 1e6))
    for k, v in acc.items():
        print("   %-13s %7.3f ms per step  (%4.1f %%)" % (k, v / n / 1e6, 100.0 * v / span))
    gaps.sort(reverse=True)
    print("   idle gaps: %d, the ten longest [us]: %s" % (len(gaps), [round(g / 1e3, 1) for g in gaps[:10]]))
    busy = {}
    for s, e, fam, name in rows:
        if s >= t0 and e <= t1 and fam != "other":
            busy[name] = busy.get(name, 0) + e - s
    print("   kernel time per step [ms] (kernels may overlap):", {k.split("::")[-1][:28]: round(v / n / 1e6, 3) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:8]})


if __name__ == "__main__":
    main()
