#!/bin/bash
# Development helper: HBM read / write bytes (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes) of the
# HBM-bound kernels (third pass: VALU / LDS / wave counters of the same kernels) of bench.py's privateCompare / msb / ReLU part at 2^24 elements; digest -> profiles/.
#   ./tools/pmc_bitwise.sh
cd /root/repo || exit 1
D=gpurun_out/pmc_bitwise
CMD="cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && rm -rf $D && mkdir -p $D && \
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/rd -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extras --size 1024 > $D/rd.log 2>&1 && \
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/wr -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extras --size 1024 > $D/wr.log 2>&1 && \
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $D/sq -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extras --size 1024 > $D/sq.log 2>&1; echo exit=\$?"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$CMD" 2>&1 | tail -40
python3 tools/pmc_bitwise_sum.py   # digest of the merged gpurun_out/pmc_bitwise -> profiles/r3_pmc_bitwise.json
