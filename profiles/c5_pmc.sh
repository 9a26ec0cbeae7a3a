#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -f csv -d $REPO/gpurun_out/pmc_c5a -o p -- python3 $REPO/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE -f csv -d $REPO/gpurun_out/pmc_c5b -o p -- python3 $REPO/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for d in ("pmc_c5a","pmc_c5b"):
  for f in glob.glob("$REPO/gpurun_out/"+d+"/**/*counter_collection.csv",recursive=True):
    per=defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].replace('ycge::','').replace('void ','').split('(')[0]
        per[(k,row["Dispatch_Id"])][row["Counter_Name"]]+=float(row["Counter_Value"])
    for (k,_),cs in per.items():
        for c,v in cs.items(): acc[k][c].append(v)
for k in acc:
    if 'false' not in k and 'taa' not in k: continue
    a={c:sum(v)/len(v) for c,v in acc[k].items()}
    print(k, "waves %.0f"%a.get('SQ_WAVES',0), "VALU instr %.1fM"%(a.get('SQ_INSTS_VALU',0)/1e6), "lane util %.2f"%(a.get('SQ_THREAD_CYCLES_VALU',0)/max(1,a.get('SQ_ACTIVE_INST_VALU',1))/64),
          "valu busy frac %.2f"%(a.get('SQ_ACTIVE_INST_VALU',0)*4/max(1,a.get('GRBM_GUI_ACTIVE',1)/8*1024)), "wait frac %.2f"%(a.get('SQ_WAIT_ANY',0)/max(1,a.get('SQ_WAVE_CYCLES',1))),
          "active frac %.2f"%(a.get('SQ_ACTIVE_INST_ANY',0)/max(1,a.get('SQ_WAVE_CYCLES',1))), "waves/simd avg %.2f"%(a.get('SQ_WAVE_CYCLES',0)*4/max(1,a.get('GRBM_GUI_ACTIVE',1)/8*1024)))
PY
