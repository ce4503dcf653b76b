#!/bin/bash
# Development helper: one rocprofv3 --pmc pass over the matmul part of bench.py; prints per-kernel averages.
#   ./tools/pmc.sh "GRBM_GUI_ACTIVE" [ABL]
cd /root/repo || exit 1
CNT="$1"; ABL="${2:-0}"
EXTRA=""; [ "$ABL" != "0" ] && EXTRA="--no-verify"
CMD="cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_tmp && MPCX_GEMM_ABLATE=$ABL timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-extras $EXTRA > gpurun_out/pmc_tmp.log 2>&1; python3 tools/pmc_print.py"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 300 -- "$CMD" 2>&1 | grep -v "^\[gpurun\]"
