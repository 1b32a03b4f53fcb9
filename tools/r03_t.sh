cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 3000 python tests/soak/n100_report.py > gpurun_out/n100.log 2>&1; tail -2 gpurun_out/n100.log | cut -c1-300; ls -la gpurun_out/r03_n100_parity.json
