#!/bin/bash
# LDS-array cycles of the Q = 1024 search kernel in three ablation modes (0 = production, 2 = no LDS-DMA, 17 = no
# fragment reads): what the DMA writes and the fragment reads each cost the LDS.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for m in 0 2 17; do
  OUT=gpurun_out/pmc_lds_$m; mkdir -p $OUT
  ISC_DEBUG_MODE=$m rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 scripts/quick_search_bench.py 10000000x1024 > $OUT/out.txt 2> $OUT/err.txt
  python3 - $OUT $m <<'PY'
import csv,glob,sys,collections
f=sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'k_dots_filter' in r['Kernel_Name']:
        agg[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
best=max(agg.values(), key=lambda c: c.get('GRBM_GUI_ACTIVE',0))
print('mode',sys.argv[2],{k:int(v) for k,v in best.items()})
PY
done
