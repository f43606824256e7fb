#!/bin/bash
# Run ON the MI355X box: FETCH_SIZE / WRITE_SIZE of tools/hbm_calib.bin (known byte counts), two separate PMC passes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-calib}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/tools/hbm_calib.bin
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o cal -- $BIN > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o cal -- $BIN > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cal -- $BIN > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
for d in fetch write; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  python tools/summarize_prof.py counter $f > $OUT/calib_${d}_by_kernel.csv
done
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python tools/summarize_prof.py trace $f > $OUT/calib_stats_by_shape.csv
rm -rf $OUT/fetch $OUT/write $OUT/stats
cat $OUT/calib_*_by_kernel.csv $OUT/calib_stats_by_shape.csv
