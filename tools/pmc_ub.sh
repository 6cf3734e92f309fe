#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_ub; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 200 $REPO/tools/ubench_hash | tail -4
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH --kernel-trace --output-format csv -d "$OUT/pmc_A" -- $REPO/tools/ubench_hash > "$OUT/A.log" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_C" -- $REPO/tools/ubench_hash > "$OUT/C.log" 2>&1
cd $REPO; python3 - <<'PY'
import csv,glob,collections
for run in ("pmc_A","pmc_C"):
    f=glob.glob("gpurun_out/prof_ub/%s/*/*_counter_collection.csv"%run)[0]
    rows=list(csv.DictReader(open(f)))
    # last 6 dispatches of interest: print per dispatch
    by=collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"],{"name":r["Kernel_Name"][:40],"grid":r["Grid_Size"],"wg":r["Workgroup_Size"],"ns":int(r["End_Timestamp"])-int(r["Start_Timestamp"])})[r["Counter_Name"]]=float(r["Counter_Value"])
    for d,v in list(by.items())[-8:]:
        print(run,d,v)
PY
