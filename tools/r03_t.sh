cd $GRAFT_REPO_ROOT; bash tools/cluster_pmc.sh r03 2>&1 | tail -3
