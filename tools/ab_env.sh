#!/bin/bash
# A/B of environment knobs on the bench workload (no torch in the process): usage tools/ab_env.sh "VAR=a" "VAR=b" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$PWD
for cfg in "$@"; do
  for rep in 1 2 3; do
    echo -n "$cfg : "; env $cfg python tools/prof_one.py free f32 4096 100 20 | tail -1
  done
done
