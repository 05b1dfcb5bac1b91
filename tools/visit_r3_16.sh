cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_c4 $OUT/prof_rs
(cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -- python $GRAFT_REPO_ROOT/tools/recipe_bench.py --resch 64 --kernel-size 3 --upsampling 256 --T 26112 --batch 8 --steps 5 > $OUT/c4_rocprof.json 2> $OUT/c4_rocprof.err); echo "c4 rc=$?"
find $OUT/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/config4_rocprofv3_kernel_stats.csv
find $OUT/prof_c4 -name "*kernel_trace.csv" -delete
(cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rs -- python $GRAFT_REPO_ROOT/tools/recipe_bench.py --steps 3 > $OUT/rs_rocprof.json 2> $OUT/rs_rocprof.err); echo "rs rc=$?"
find $OUT/prof_rs -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/recipe_size_rocprofv3_kernel_stats.csv
find $OUT/prof_rs -name "*kernel_trace.csv" -delete
head -8 $OUT/config4_rocprofv3_kernel_stats.csv; head -8 $OUT/recipe_size_rocprofv3_kernel_stats.csv
