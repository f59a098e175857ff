"""torch-free driver for PMC collection (rocprofv3 --pmc segfaults with the torch-bundled HIP runtime in the process):
B sequences x a few frames through the host-image entry point."""
import os, sys
os.environ["GF_NO_TORCH_PRELOAD"] = "1"
sys.path.insert(0, "ground-fusion_amd")
import numpy as np, gfamd, synth
B, N = 32, 4
seqs = [synth.tracker_sequence(1000 + b, N) for b in range(B)]
trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=B))
for k in range(N):
    trk.trackImageBatch([k / 15.0] * B, [seqs[b][k] for b in range(B)], None)
st = trk.stats()
print("lk_launches", st["lk_launches"], "alg_bytes_per_launch", (484.0 * 5 * st["lk_level_passes"] + 484.0 * st["lk_iterations"]) / max(st["lk_launches"], 1))
trk.close()
