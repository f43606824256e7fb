#!/bin/bash
# Run ON the MI355X box: LDS-array counters of every kernel of the config-2 step (eager launches, one --pmc pass, --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c2_lds}
mkdir -p $O
export ANODDPM_NO_GRAPH=1
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --no-extra ${2:-}"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o c2 -- $B > $O/sq.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py counter $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/c2_lds_by_kernel.csv
rm -rf $O/sq
head -5 $O/c2_lds_by_kernel.csv
