cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o pf -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | cut -c1-200
f=$(find /tmp/pf -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-150
