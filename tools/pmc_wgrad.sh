#!/bin/bash
# Run ON the MI355X box: SQ counters of the 3x3 weight-gradient kernels on the layer shapes of config 3 (tools/bench_conv.py WG=1).
# $1 = output tag.  Two passes: the eight SQ counters, then the LDS counters (counters never share a run with other trace domains).
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-wgrad_pmc}
mkdir -p $O
WG=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o wg -- python $GRAFT_REPO_ROOT/tools/bench_conv.py > $O/sq.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/wgrad_sq_by_kernel.csv
python $GRAFT_REPO_ROOT/tools/summarize_prof.py trace $(find $O/sq -name "*kernel_trace.csv" | head -1) > $O/wgrad_by_shape.csv
rm -rf $O/sq
WG=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM --output-format csv -d $O/lds -o wg -- python $GRAFT_REPO_ROOT/tools/bench_conv.py > $O/lds.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/lds -name "*counter_collection.csv" | head -1) > $O/wgrad_lds_by_kernel.csv
rm -rf $O/lds
grep -i "wgrad43" $O/wgrad_sq_by_kernel.csv $O/wgrad_lds_by_kernel.csv | head -60
grep "wgrad" $O/sq.log | head -10
