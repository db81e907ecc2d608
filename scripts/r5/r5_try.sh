#!/bin/bash
# round 5: correctness of the working copy's suffix sort on both initial sorts + smoke, then the quick look (r5_base.sh)
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py msd 2>&1 | tail -4
python scripts/gpu_msd_check.py lsd 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r5/r5_base.sh ${1:-try}
