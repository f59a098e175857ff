#!/bin/bash
# round 6, closing run: the measurement set of the record on the final tree, then the whole GPU suite once more with every alternative form switched on
cd "$(dirname "$0")/.." || exit 1
bash scripts/r06_run.sh r06_zz tests bench1 bench2 bench4 driver prof1 backend tracker pmc
GF_LK_POINTS=4 GF_BA_CHAIN=1 timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_zz_pytest_alternatives.log 2>&1
echo "alternatives (GF_LK_POINTS=4 GF_BA_CHAIN=1):"; tail -8 gpurun_out/r06_zz_pytest_alternatives.log | cut -c1-200
