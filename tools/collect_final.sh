#!/bin/bash
# Run ON the MI355X box: the end-of-round subset of collect_profiles.sh (bench lines of all configs + the c3 kernel statistics).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --config c3 --steps 5 --warmup 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --config c5 --steps 10 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
cd /tmp && export TMPDIR=/tmp
export ANODDPM_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --config c3 > $OUT/c3_stats.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/c3_stats -name "*kernel_trace.csv"); do python tools/summarize_prof.py trace $f > $OUT/c3_stats_by_shape.csv; done
for f in $(find $OUT/c3_stats -name "*kernel_stats.csv"); do cp $f $OUT/c3_stats_kernel_stats.csv; done
rm -rf $OUT/c3_stats
ls $OUT
