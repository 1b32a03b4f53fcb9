#!/bin/bash
# Shared backward sweep A/B on one box: DIRECT_DDP_BSHARE = 0 (owner-only sweeps) / 1 (helpers) on the fixed-20 launch of
# config 2 at several batch sizes.  usage (through gpurun): bash tools/bshare_ab.sh [batches...]
cd ${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$PWD
for B in ${@:-4096 1 256 3072 16384}; do
  for rep in 1 2; do
    for m in 0 1; do
      echo -n "BSHARE=$m "; DIRECT_DDP_BSHARE=$m timeout 300 python tools/ab_time.py free f32 5 $B | tail -1
    done
  done
done
