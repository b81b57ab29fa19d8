#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy; each in its own run, eager launches) for BASELINE configs[3] (batch 64)
# and configs[4] (256 x 256, batch 16): VERDICT r4 item 7.  One collect_evidence.sh run per configuration into gpurun_out/ev_<tag>;
# EV_OUT=ev_<tag> python3 scratch/summarise_profiles.py r05_<tag> then writes profiles/r05_<tag>_pmc_{traffic,mfma}.json.
R=$(cd "$(dirname "$0")/.." && pwd)
EV_FLAGS="--batch-per-gpu 64 --dtype bf16 --no-cpu-baseline" EV_OUT=ev_c3_bf16 bash $R/scratch/collect_evidence.sh || exit 1
EV_FLAGS="--size 256 --batch-per-gpu 16 --dtype bf16 --no-cpu-baseline" EV_OUT=ev_c4_bf16 bash $R/scratch/collect_evidence.sh || exit 1
EV_FLAGS="--size 256 --batch-per-gpu 16 --no-cpu-baseline" EV_OUT=ev_c4_fp32 bash $R/scratch/collect_evidence.sh || exit 1
