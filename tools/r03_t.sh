cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
DIRECT_DDP_LIB=$PWD/build_variants/cluster_counts.so timeout 600 python tools/cluster_counts.py 2>&1 | tail -2
