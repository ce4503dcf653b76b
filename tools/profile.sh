#!/bin/bash
# Development helper: rocprofv3 kernel trace of the default bench.py run plus two PMC passes (HBM read / write
# bytes) over its matmul part; results land in gpurun_out/prof_<tag>/ and are summarised by tools/profile_sum.py.
#   ./tools/profile.sh r1b
cd /root/repo || exit 1
TAG="${1:-tmp}"
D=gpurun_out/prof_$TAG
CMD="cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && rm -rf $D && mkdir -p $D && \
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python3 bench.py --no-cpu > $D/bench_trace.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_rd -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-extras > $D/pmc_rd.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_wr -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-extras > $D/pmc_wr.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_sq -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-extras > $D/pmc_sq.log 2>&1; echo exit=\$?; ls $D; python3 tools/profile_sum.py $D"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$CMD" 2>&1 | tail -60
