# (gpurun) everything profiles/r04_delaunay_* is made of
O=gpurun_out/r04_dt; mkdir -p $O
timeout 300 python tools/exp/delaunay_gpu_time.py > $O/time.txt 2>&1
timeout 400 bash tools/exp/delaunay_prof.sh > $O/per_config.txt 2>&1
timeout 600 bash tools/exp/delaunay_pmc.sh > $O/pmc.txt 2>&1
# (work per star: needs delaunay_dev.hip compiled with -DFLAME_DT_STATS=1 -- FLAME_EXTRA_HIPCC_FLAGS=-DFLAME_DT_STATS=1 python flame_ros_amd/build.py --force)
timeout 300 python tools/exp/delaunay_stats.py 2>&1 | grep "\[dt\]" > $O/work.txt
timeout 600 python tools/exp/from_features.py 2>&1 | grep triangulation > $O/from_features.txt
tail -n 20 $O/time.txt $O/per_config.txt $O/pmc.txt $O/work.txt $O/from_features.txt
