cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 3000 python tests/soak/parity_soak.py 36 > gpurun_out/r03_parity_soak.log 2>&1; tail -3 gpurun_out/r03_parity_soak.log
timeout 600 python tests/soak/help_stress.py > gpurun_out/r03_help_stress.log 2>&1; tail -3 gpurun_out/r03_help_stress.log
