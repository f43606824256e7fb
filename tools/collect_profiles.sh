#!/bin/bash
# Run ON the MI355X box (gpurun): collects the rocprofv3 passes behind profiles/r<round>_* into gpurun_out/$1/.
# Counters are collected in their own passes with --kernel-trace only (no other trace domains).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --no-extra"
export ANODDPM_NO_GRAPH=1            # eager launches so that every kernel is attributed by name
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -o c2 -- $B > $OUT/c2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -o c2 -- $B > $OUT/c2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c2_write -o c2 -- $B > $OUT/c2_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/c2_sq -o c2 -- $B > $OUT/c2_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- $B --config c3 --steps 3 --warmup 1 > $OUT/c3_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o c4 -- $B --config c4 > $OUT/c4_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/c4_sq -o c4 -- $B --config c4 > $OUT/c4_sq.log 2>&1
unset ANODDPM_NO_GRAPH
cd $GRAFT_REPO_ROOT
# per-layer profile from the executor's own HIP events (round 4: bench.py --dump-layers; split-K tails are inside their layer's time)
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --dump-layers $OUT/c2_igemm_by_layer.csv > /dev/null 2> $OUT/by_layer.err
# one step as a timeline (kernel, start, duration) from the eager trace
python tools/timeline.py $(find $OUT/c2_stats -name "*kernel_trace.csv" | head -1) -2 $OUT/c2_step_timeline.csv > $OUT/c2_step_timeline_summary.txt 2>> $OUT/by_layer.err
# keep what is small: summaries only (the raw traces can be hundreds of MB)
for d in c2_stats c2_fetch c2_write c2_sq c3_stats c4_stats c4_sq; do
  for f in $(find $OUT/$d -name "*kernel_trace.csv" -o -name "*counter_collection.csv" 2>/dev/null); do
    case $f in
      *kernel_trace.csv) python tools/summarize_prof.py trace $f > $OUT/${d}_by_shape.csv ;;
      *counter_collection.csv) python tools/summarize_prof.py counter $f > $OUT/${d}_by_kernel.csv ;;
    esac
  done
  for f in $(find $OUT/$d -name "*kernel_stats.csv" 2>/dev/null); do cp $f $OUT/${d}_kernel_stats.csv; done
  rm -rf $OUT/$d
done
# HBM-side bytes per launch for bench.py's roofline.traffic (calibration: tools/calib_hbm.sh -> profiles/r3_hbm_calibration.json)
python tools/traffic_from_pmc.py traffic profiles/r3_hbm_calibration.json $OUT/c2_fetch_by_kernel.csv $OUT/c2_write_by_kernel.csv 4 "${2:-}" > $OUT/traffic_c2.json
# the bench lines last: roofline.traffic of the c2 line reads the traffic file of THIS build (committed later under the same name)
cp $OUT/traffic_c2.json profiles/${1:-prof}_traffic_c2.json
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --config c3 --steps 5 --warmup 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --config c4 --steps 10 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --config c5 --steps 10 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python bench.py --config det --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_det.json 2> $OUT/bench_det.err   # a step = one image's whole sweep (~35 s)
ls -la $OUT
