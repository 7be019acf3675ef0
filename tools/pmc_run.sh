#!/bin/bash
# rocprofv3 PMC passes (one counter set per run, no trace domains mixed in) of one python command:
#   tools/pmc_run.sh <tag> <script> [args..]   -> gpurun_out/pmc_<tag>/summary.txt
# per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KB) and the HBM bytes per launch = 2 * FETCH + WRITE (the gfx950
# correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 64 B per 128-B request of a wide streaming read)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r set; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/"$@" </dev/null > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<SETS
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH
SETS
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::native" in k or "rocprim" in k or "elementwise" in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in agg.items():
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    fm, wm = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
    rows.append((2 * fm + wm, k, len(f), fm, wm, d))
with open("$OUT/summary.txt", "w") as o:
    for hb, k, n, fm, wm, d in sorted(rows, reverse=True):
        o.write(f"{k[:70]:70s} launches {n:5d}  FETCH_SIZE {fm:12.0f} KB  WRITE_SIZE {wm:12.0f} KB  => HBM bytes per launch (2*FETCH + WRITE) {hb * 1024 / 1e9:8.4f} GB\n")
        for c in sorted(d):
            if c in ("FETCH_SIZE", "WRITE_SIZE"): continue
            o.write(f"      {c:28s} mean {sum(d[c]) / len(d[c]):.6g}\n")
print(open("$OUT/summary.txt").read()[:6000])
PY
