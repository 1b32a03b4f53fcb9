bash tools/scratch/pmc.sh
bash tools/scratch/pmc2.sh
cat gpurun_out/pmc/report[1-4].txt gpurun_out/pmc/report1[1-3].txt | grep "k_iterate" > gpurun_out/pmc/all.txt
