"""end-to-end drop-in path (bench.py's end_to_end_sample) at several group sizes: python scripts/e2e_bench.py 64 256"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
import torch, gfamd, bench
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [64]:
    r = bench.end_to_end_sample(gfamd, n, dev, 150, 30)
    print({k: v for k, v in r.items() if k != "path"}, flush=True)
