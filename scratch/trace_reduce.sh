#!/bin/bash
# kernel-trace of a short bench run: what do the slab-sum kernels cost with and without srgan_wgrad_defer?
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/trred
cd /tmp && export TMPDIR=/tmp
for m in defer immediate; do
  if [ $m = immediate ]; then export SRGAN_NO_WGRAD_DEFER=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trred/$m -- python3 $R/bench.py --steps 4 --warmup 2 --graph off --no-cpu-baseline --no-micro > /dev/null 2> $R/gpurun_out/trred/$m.err
  f=$(ls $R/gpurun_out/trred/$m/*/*kernel_stats.csv | head -1)
  echo "== $m"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows:
    if "reduce" in r["Name"] or "colsum" in r["Name"]:
        print("%-70s calls %5s avg %8.1f us total %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
