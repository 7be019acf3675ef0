#!/bin/bash
# usage: tools/prof_stats.sh <tag> <env assignments...> ; kernel-trace stats of tools/probe_only.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$TAG -- python3 $R/tools/probe_only.py > $R/gpurun_out/st_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/st_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = open("$R/gpurun_out/st_$TAG.txt", "w")
for r in rows:
    n = r["Name"]
    if "at::" in n or "rocclr" in n or "rocprim" in n: continue
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    line = f'{n.split("(")[0][-48:]:50s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:10.1f} min_us={float(r["MinNs"])/1e3:10.1f} max_us={float(r["MaxNs"])/1e3:10.1f}'
    print(line); out.write(line + "\n")
PY
