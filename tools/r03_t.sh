cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for rep in 1 2; do for lib in libdirect_ddp.so ab_norowcache.so; do
  echo -n "$lib: "; DIRECT_DDP_LIB=$PWD/direct_amd/lib/$lib python tools/ab_time.py free f32 5 4096 | tail -1 | cut -c1-150
  echo -n "$lib: "; DIRECT_DDP_LIB=$PWD/direct_amd/lib/$lib python tools/ab_time.py free f32 3 16384 | tail -1 | cut -c1-150
done; done
