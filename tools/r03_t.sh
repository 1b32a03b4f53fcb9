cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
bash tools/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1; tail -2 gpurun_out/r03_profile.log | cut -c1-200
head -4 gpurun_out/r03_bench_kernel_stats.csv | cut -c1-150
cut -c1-300 gpurun_out/r03_bench.json
