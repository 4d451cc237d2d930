#!/bin/bash
# GPU box: rocprofv3 evidence for the round (run from the repo root through gpurun).  Outputs under gpurun_out/prof_r03*/;
# the summaries that are judged are copied into profiles/ afterwards (tools/make_traffic.py, profiles/README.md).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --no-ppo"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03 -o r03 --output-format csv -- $B --steps 5 --warmup 2 > $R/gpurun_out/prof_r03.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r03_fetch -o f --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_r03_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r03_write -o w --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_r03_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_r03_sq -o s --output-format csv -- $B --steps 1 --warmup 1 > $R/gpurun_out/prof_r03_sq.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_r03_clk -o c --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_r03_clk.log 2>&1 || exit 1
cd $R && python3 tools/effective_clock.py gpurun_out/prof_r03_clk > gpurun_out/prof_r03_clock.txt 2>&1
cd $R && python3 tools/make_traffic.py gpurun_out/prof_r03_fetch gpurun_out/prof_r03_write 512 > gpurun_out/prof_r03_traffic.txt 2>&1
cp profiles/traffic.json gpurun_out/r03_traffic.json
python3 tools/pmc_summary.py gpurun_out/prof_r03_sq cwlt > gpurun_out/prof_r03_sq_summary.txt 2>&1
echo done
