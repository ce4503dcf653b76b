#!/bin/bash
# Diagnostic build of the library (-DMPCX_DEV_PROBES: timing ablations of the digit-pair GEMM, clock stamps in the
# residue GEMM).  Never loaded by the product, the tests or bench.py; tools/gemm_clock_probe.py loads it by path.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-result -DMPCX_DEV_PROBES \
    -o tools/libmpcx_probes.so mpc_amd/csrc/{ctx,prg,ring_gemm,ring_gemm_rns,aby3,securenn,myengine,group}.hip
echo built tools/libmpcx_probes.so
