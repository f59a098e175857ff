import sys, time
sys.path.insert(0,'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est=gfamd.Estimator(batch=256)
base=[SW.make_window(1000+b, gfamd) for b in range(8)]
wins=[base[b%8].copy() for b in range(256)]
for rep in range(3):
    ws=[w.copy() for w in wins]
    est.reset_stats()
    t0=time.perf_counter(); est.solve(ws, 8); t1=time.perf_counter()
    print('solve 256 windows: wall %.2f ms' % (1e3*(t1-t0)), {k: round(v,3) for k,v in est.stats().items() if k.startswith('ms_')})
