#!/bin/bash
# SQ counters of the attention kernels at one shape (default: 32 heads-of-8 x 4096 tokens, d = 40): issue mix and how much
# of the matrix pipe's busy time overlaps vector issue.  usage: tools/pmc_attn.sh "B H N D[ T]"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_pmc
rm -rf $O; mkdir -p $O
shape=${1:-32 8 4096 40}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/p1 -o p -- python3 $R/tools/attn_bench.py $shape > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- python3 $R/tools/attn_bench.py $shape > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
find $O -name '*.db' -delete
python3 - <<PY
import csv, collections
for p in ('p1', 'p2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open('$O/%s/p_counter_collection.csv' % p)):
        if 'attn' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(p, k, {c: '%.4g' % (sum(v[3:]) / len(v[3:])) for c, v in d.items()})
PY
