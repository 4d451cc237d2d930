#!/bin/bash
# GPU box: rocprofv3 evidence for the round (run from the repo root through gpurun: `bash tools/profile_round.sh r04`).  Outputs under gpurun_out/prof_<tag>*/;
# the summaries that are judged are copied into profiles/ afterwards (tools/make_traffic.py, profiles/README.md).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --no-ppo"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T} -o ${T} --output-format csv -- $B --steps 5 --warmup 2 > $R/gpurun_out/prof_${T}.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${T}_fetch -o f --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_${T}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${T}_write -o w --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_${T}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_${T}_sq -o s --output-format csv -- $B --steps 1 --warmup 1 > $R/gpurun_out/prof_${T}_sq.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_${T}_clk -o c --output-format csv -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_${T}_clk.log 2>&1 || exit 1
cd $R && python3 tools/effective_clock.py gpurun_out/prof_${T}_clk > gpurun_out/prof_${T}_clock.txt 2>&1
cd $R && python3 tools/make_traffic.py gpurun_out/prof_${T}_fetch gpurun_out/prof_${T}_write 512 > gpurun_out/prof_${T}_traffic.txt 2>&1
cp profiles/traffic.json gpurun_out/${T}_traffic.json
python3 tools/pmc_summary.py gpurun_out/prof_${T}_sq cwlt > gpurun_out/prof_${T}_sq_summary.txt 2>&1
echo done
