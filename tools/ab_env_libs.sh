#!/bin/bash
# A/B of library builds on ONE box, optionally with an environment setting per entry: usage tools/ab_env_libs.sh "lib.so[,VAR=1]" ...
cd ${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$PWD
run() { local spec=$1 B=$2; local lib=${spec%%,*}; local envs=""; [ "$lib" != "$spec" ] && envs=${spec#*,}
  echo -n "$spec B=$B: "; env $envs DIRECT_DDP_LIB=$PWD/$lib python tools/prof_one.py free f32 $B 100 20 | tail -1; }
for rep in 1 2 3; do for s in "$@"; do run $s 4096; done; done
[ -n "$AB_SKIP_BIG" ] || for rep in 1 2; do for s in "$@"; do run $s 16384; done; done
