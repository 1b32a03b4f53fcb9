cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python $GRAFT_REPO_ROOT/tests/soak/cluster_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/cb.err
tail -3 $GRAFT_REPO_ROOT/gpurun_out/cb.err; cat $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_bench.json | tail -1 | cut -c1-1800
f=$(find /tmp/pc -name '*kernel_stats.csv' | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_kernel_stats.csv; head -14 $f | cut -c1-150
