cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 1500 python tests/soak/real_corridor_bench.py > gpurun_out/r03_real_corridors.json 2> gpurun_out/rc.err; tail -2 gpurun_out/rc.err; cut -c1-1500 gpurun_out/r03_real_corridors.json
