#!/bin/bash
# EV_FLAGS (optional, e.g. "--dtype bf16"): extra bench.py flags for every pass.  EV_OUT (optional): directory under gpurun_out/ (default ev).
# One evidence run on the GPU box: un-profiled bench (with the CPU baseline and the micro-benchmark), --stats run,
# FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (each in its own run, eager launches: same kernels as the graph replays).
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/${EV_OUT:-ev}
rm -rf $E && mkdir -p $E
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $EV_FLAGS --steps 20 --warmup 5 > $E/bench_plain.json 2> $E/bench_plain.err || exit 1
echo "plain done"
python3 $R/bench.py $EV_FLAGS --steps 20 --warmup 5 --graph off --no-cpu-baseline --no-micro > $E/bench_eager.json 2> $E/bench_eager.err || exit 1
echo "eager done"
rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -- python3 $R/bench.py $EV_FLAGS --steps 5 --warmup 2 --no-cpu-baseline --no-micro > $E/bench_under_rocprof.json 2> $E/stats.err || exit 1
echo "stats done"
# Round 6: the counter passes run SERIALISED (AMD_SERIALIZE_KERNEL=3) and under a time limit.  An eager bench.py enqueues every step
# without synchronising; past ~8000 outstanding dispatches rocprofv3's counter-collecting queue aborts with
# HSA_STATUS_ERROR_INVALID_PACKET_FORMAT and the process then never exits (profiles/LOG.md, round 6).  Counters are per dispatch.
export AMD_SERIALIZE_KERNEL=3
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/fetch -- python3 $R/bench.py $EV_FLAGS --steps 5 --warmup 1 --graph off --no-cpu-baseline --no-micro > /dev/null 2> $E/fetch.err || exit 1
echo "fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/write -- python3 $R/bench.py $EV_FLAGS --steps 5 --warmup 1 --graph off --no-cpu-baseline --no-micro > /dev/null 2> $E/write.err || exit 1
echo "write done"
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $E/mfma -- python3 $R/bench.py $EV_FLAGS --steps 5 --warmup 1 --graph off --no-cpu-baseline --no-micro > /dev/null 2> $E/mfma.err || exit 1
echo "mfma done"
unset AMD_SERIALIZE_KERNEL
# keep the merge-back small: counter CSVs only (traces of the PMC passes are large)
find $E/fetch $E/write $E/mfma -name "*kernel_trace.csv" -delete
find $E/stats -name "*kernel_trace.csv" -delete
