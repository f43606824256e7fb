#!/bin/bash
# Run ON the MI355X box: SQ counters of the F(4x4,3x3) kernels on the layer shapes of config 2 (tools/bench_conv.py ONLY3=1 FULL=1).
# $1 = output tag.  One pass, eight SQ counters, --kernel-trace only (counters never share a run with other trace domains).
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-f43_pmc}
mkdir -p $O
ONLY3=1 FULL=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o f43 -- python $GRAFT_REPO_ROOT/tools/bench_conv.py > $O/sq.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/f43_sq_by_kernel.csv
python $GRAFT_REPO_ROOT/tools/summarize_prof.py trace $(find $O/sq -name "*kernel_trace.csv" | head -1) > $O/f43_by_shape.csv
rm -rf $O/sq
ONLY3=1 FULL=1 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -o f43 -- python $GRAFT_REPO_ROOT/tools/bench_conv.py > $O/sq2.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq2 -name "*counter_collection.csv" | head -1) > $O/f43_sq2_by_kernel.csv
rm -rf $O/sq2
grep "wino43r" $O/f43_sq_by_kernel.csv $O/f43_sq2_by_kernel.csv | head -60
grep "^B" $O/sq.log | cut -c1-80
