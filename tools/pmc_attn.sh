#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_pmc
rm -rf $O; mkdir -p $O
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p1 -o p -- python3 $R/tools/attn_bench.py 32 8 1024 40 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- python3 $R/tools/attn_bench.py 32 8 1024 40 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
find $O -name '*.db' -delete
ls -R $O | head
