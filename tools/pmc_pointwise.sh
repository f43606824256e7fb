cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pw
mkdir -p $O
PW=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o pw -- python $GRAFT_REPO_ROOT/tools/bench_conv.py > $O/sq.log 2>&1
f=$(find $O/sq -name "*counter_collection.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $f > $O/sq_by_kernel.csv
g=$(find $O/sq -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/summarize_prof.py trace $g > $O/sq_by_shape.csv
rm -rf $O/sq
grep -i "pointwise\|igemm_kernel" $O/sq_by_kernel.csv | head -60
grep -i "pointwise\|igemm_kernel" $O/sq_by_shape.csv | head
tail -3 $O/sq.log
