#!/bin/bash
# A/B of library builds on ONE box (box-to-box variance is a few percent): usage tools/ab_libs.sh a.so b.so ...
cd ${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$PWD
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib) B=4096: "; DIRECT_DDP_LIB=$PWD/$lib python tools/prof_one.py free f32 4096 100 20 | tail -1
  done
done
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) B=16384: "; DIRECT_DDP_LIB=$PWD/$lib python tools/prof_one.py free f32 16384 100 20 | tail -1
  done
done
