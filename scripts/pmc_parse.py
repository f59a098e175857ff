"""per-kernel averages of a rocprofv3 counter_collection.csv: kernel, counter, launches, mean value per launch, min, max, mean kernel duration [ns]"""
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if not (n.startswith(("gf", "void gf", "calib_"))):
        continue
    key = n.split("(")[0].replace("void ", "")
    acc[key][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    dur[key][int(r.get("Dispatch_Id", 0))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernel,counter,launches,mean_per_launch,min,max,mean_duration_ns")
for k in sorted(acc):
    for c in sorted(acc[k]):
        per = collections.defaultdict(float)   # one dispatch may be reported in several rows: sum rows of the same dispatch
        for d, v in acc[k][c]:
            per[d] += v
        vals = list(per.values())
        print("%s,%s,%d,%.1f,%.1f,%.1f,%.0f" % (k.replace(",", ";"), c, len(vals), sum(vals) / len(vals), min(vals), max(vals), sum(dur[k].values()) / len(dur[k])))
