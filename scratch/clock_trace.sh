#!/bin/bash
# Shader clock and socket power while the bench replays its graph (is the step power / clock limited?).  Samples rocm-smi every
# 0.5 s next to `bench.py "$@"`; output: gpurun_out/clock_<tag>.txt (samples) and clock_<tag>.json (the bench line).
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
O=$R/gpurun_out
mkdir -p $O
rocm-smi --showclocks --showpower > $O/clock_${tag}_idle.txt 2>&1
python3 $R/bench.py --no-cpu-baseline --no-micro "$@" > $O/clock_$tag.json 2> $O/clock_$tag.err &
pid=$!
: > $O/clock_$tag.txt
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power"; } >> $O/clock_$tag.txt
  sleep 0.5
done
wait $pid
