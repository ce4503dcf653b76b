#!/bin/bash
# Development helper: time the matmul bench and collect FETCH_SIZE for several tile-group heights.
cd /root/repo || exit 1
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i -A5 "error"
CMD="cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && mkdir -p gpurun_out"
for GM in $@; do
CMD="$CMD && echo GM=$GM && MPCX_GEMM_GM=$GM timeout -k 10 120 python3 -u bench.py --no-bitwise --no-cpu --steps 3 --warmup 1 > gpurun_out/probe.log 2>&1 && python3 tools/probe_print.py && rm -rf gpurun_out/pmc_tmp && MPCX_GEMM_GM=$GM timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise > gpurun_out/pmc_tmp.log 2>&1 && python3 tools/pmc_print.py"
done
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$CMD" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)"
