cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
bash tools/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1; tail -2 gpurun_out/r03_profile.log | cut -c1-200
for c in "r03_config3 corridor f32 4096 100 20" "r03_config4 corridor f64 16384 300 20" "r03_config5 corridor f32 16384 100 20"; do bash tools/pmc_config.sh $c 2>&1 | tail -1 | cut -c1-200; done
bash tools/configs_round.sh r03; cat gpurun_out/r03_cluster_bench.json | cut -c1-300
