#!/bin/bash
# the config-A step with every grouped weight-gradient launch forced onto one tile (SKYEMB_GROUP_TILE), interleaved with the plan's choice
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for t in 0 128128 128064 64064 9128128; do
    if [ $t = 0 ]; then unset SKYEMB_GROUP_TILE; else export SKYEMB_GROUP_TILE=$t; fi
    echo -n "tile $t: "; timeout -k 10 120 python tools/step_only.py 2>&1 | grep "ms/step"
  done
done
