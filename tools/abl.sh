#!/bin/bash
# Development helper: time ablated GEMM variants (no result check) and the clock they hold.
cd /root/repo || exit 1
CMD="cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && mkdir -p gpurun_out"
for A in $@; do
CMD="$CMD && echo ABL=$A && MPCX_GEMM_ABLATE=$A timeout -k 10 120 python3 -u bench.py --no-bitwise --no-cpu --no-verify --steps 3 --warmup 1 > gpurun_out/probe.log 2>&1 && python3 tools/probe_print.py && rm -rf gpurun_out/pmc_tmp && MPCX_GEMM_ABLATE=$A timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-verify > gpurun_out/pmc_tmp.log 2>&1 && python3 tools/pmc_print.py"
done
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$CMD" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | grep -v prg_fill
