#!/bin/bash
# PMC passes over tools/probe_only.py (one counter set per run; no trace domains mixed in)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$1
shift
[ $# -gt 0 ] && export "$@"
mkdir -p $OUT
i=0
while read -r set; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/probe_only.py </dev/null > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<SETS
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
SETS
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_probe" not in k and "k_part" not in k and "k_fill" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"  {c:36s} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
PY
