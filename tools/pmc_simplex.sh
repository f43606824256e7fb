#!/bin/bash
# Run ON the MI355X box: SQ counters of the config-4 simplex launch (one pass, --kernel-trace only).  $1 = output tag, $2 = library tag
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c4_pmc}
mkdir -p $O
export ANODDPM_LIB_TAG=${2:-}
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-prof --no-extra"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o c4 -- $B > $O/sq.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq -name "*counter_collection.csv" | head -1) | grep "^kernel\|simplex" > $O/c4_sq_by_kernel.csv
rm -rf $O/sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/sq2 -o c4 -- $B > $O/sq2.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq2 -name "*counter_collection.csv" | head -1) | grep "^kernel\|simplex" > $O/c4_sq2_by_kernel.csv
rm -rf $O/sq2
cat $O/c4_sq_by_kernel.csv $O/c4_sq2_by_kernel.csv | cut -c1-160
tail -3 $O/sq2.log
