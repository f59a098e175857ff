"""per-launch durations of the back-end kernels by grid size (run under rocprofv3 --kernel-trace --output-format csv; parse with --parse DIR)"""
import sys, glob, csv, collections
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    acc = collections.defaultdict(list)
    for fn in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            n = r["Kernel_Name"]
            if "gfb::" in n:
                key = (n.split("(")[0].replace("void ", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
                acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print("%-40s grid %-8s wg %-4s calls %4d  median %8.1f us  p90 %8.1f us  max %8.1f us  total %8.1f ms" % (k[0], k[1], k[2], len(v), v[len(v) // 2] / 1e3, v[len(v) * 9 // 10] / 1e3, v[-1] / 1e3, sum(v) / 1e6))
    sys.exit(0)
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
B = 256
est = gfamd.Estimator(batch=B)
base = [SW.make_window(1000 + b, gfamd) for b in range(16)]
wins = [base[b % 16] for b in range(B)]
est.upload(wins)
for it in range(5):
    est.solve_resident(8, 0, True)
