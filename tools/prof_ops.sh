#!/bin/bash
# kernel-trace stats of tools/ops_perf.py: tools/prof_ops.sh <tag> OPS=merge [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/po_$TAG -- python3 $R/tools/ops_perf.py > $R/gpurun_out/po_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/po_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = open("$R/gpurun_out/po_$TAG.txt", "w")
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    line = f'{n.split("(")[0][-52:]:54s} calls={r["Calls"]:>5s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:10.1f} pct={r["Percentage"]}'
    print(line); out.write(line + "\n")
PY
