cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 2400 python tests/soak/parity_soak.py > gpurun_out/soak.log 2>&1; tail -3 gpurun_out/soak.log | cut -c1-600
timeout 1500 python tests/soak/help_stress.py > gpurun_out/help_stress.log 2>&1; tail -2 gpurun_out/help_stress.log | cut -c1-400
