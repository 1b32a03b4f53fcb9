cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 1500 python tests/soak/hull_soak.py 20 > gpurun_out/r03_hull_soak.json 2> gpurun_out/hs.err; tail -3 gpurun_out/hs.err; cut -c1-900 gpurun_out/r03_hull_soak.json
