#!/bin/bash
# rocprofv3 kernel stats of scratch/bench_conv.py for single layers: prof_conv.sh <tag> <ONLY substring> [env...]
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; only=$2; shift 2
O=$R/gpurun_out/pc_$tag
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ONLY="$only" "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scratch/bench_conv.py > $O/out.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "== $tag ($only) =="; grep -v "^/opt" $O/out.txt | tail -3
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-80s x%4d avg %8.1f us"%(r["Name"][:80].replace("srgan::",""), int(r["Calls"]), float(r["AverageNs"])/1e3))
PY
find $O -name "*kernel_trace.csv" -delete
