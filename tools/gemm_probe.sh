#!/bin/bash
# Development helper: rebuild the library, run the matmul part of bench.py on the GPU box, print the GEMM timing.
#   ABL=1|2 ./tools/gemm_probe.sh   -> ablated kernel variants (no result check)
cd /root/repo || exit 1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i -A5 "error"
EXTRA=""
[ -n "$ABL" ] && EXTRA="--no-verify"
CMD="mkdir -p gpurun_out && MPCX_GEMM_ABLATE=${ABL:-0} timeout -k 10 300 python -u bench.py --no-bitwise --no-cpu --steps 3 --warmup 1 $EXTRA > gpurun_out/probe.log 2>&1; echo exit=\$?; python3 tools/probe_print.py"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 600 -- "$CMD" 2>&1 | tail -4
