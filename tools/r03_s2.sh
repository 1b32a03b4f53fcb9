#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT; export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py tests/test_gpu_configs.py -x -q > $OUT/r03_s2_pytest.log 2>&1
tail -15 $OUT/r03_s2_pytest.log
bash tools/quick_perf.sh > $OUT/r03_s2_perf.log 2>&1
cat $OUT/r03_s2_perf.log
timeout 1500 python tests/soak/n100_report.py 64 > $OUT/r03_n100_report.log 2>&1
tail -c 600 $OUT/r03_n100_report.log
