cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_hull.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python $GRAFT_REPO_ROOT/tests/soak/cluster_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/cb.err
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_bench.json | cut -c1-330;  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_bench.json | grep -o '"seeds_to_planes_chain.*'
f=$(find /tmp/pc -name '*kernel_stats.csv' | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03_cluster_kernel_stats.csv; head -5 $f | cut -c1-120
