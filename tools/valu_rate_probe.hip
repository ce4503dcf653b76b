// Development probe (stand-alone: hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe).
// What the vector ALU of one SIMD sustains, in wave instructions per shader cycle, for the instruction kinds the
// generator kernels are made of -- the denominator of "VALU busy" for pc_party_gen_kernel / bit_share_gen_kernel /
// prg_mod67_kernel, whose counters give wave instructions and cycles but no busy fraction on gfx950.
//   mode 0: 32-bit adds (8 independent chains per lane)      mode 1: 32-bit low multiplies (v_mul_lo_u32)
//   mode 2: 64-bit adds (v_add_co / v_addc pairs)            mode 3: prvhash_core64 steps (utils/PRgenerator.cpp:3-19 of the
//                                                                    reference; two independent generators per lane)
// Every CU runs `waves` waves per SIMD; a lane's result is stored once at the end.  Clock: s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint64_t u64;
typedef uint32_t u32;

__device__ inline u64 prv_step(u64 &seed, u64 &lcg, u64 &hash) {
    seed *= lcg * 2 + 1;
    asm volatile("" : "+v"(seed));
    const u64 rs = seed >> 32 | seed << 32;
    hash += rs + 0xAAAAAAAAAAAAAAAAull;
    lcg += seed + 0x5555555555555555ull;
    seed ^= hash;
    return lcg ^ rs;
}

template <int MODE> __global__ __launch_bounds__(256) void k(u64 *out, u64 *stamps, int iters) {
    const u32 tid = blockIdx.x * 256 + threadIdx.x;
    u32 a[8];
    u64 b[4];
    for (int i = 0; i < 8; i++) a[i] = tid * 2654435761u + i;
    for (int i = 0; i < 4; i++) b[i] = (u64)tid * 0x9E3779B97F4A7C15ull + i;
    u64 s0 = b[0], l0 = b[1], h0 = 0, s1 = b[2], l1 = b[3], h1 = 0, acc = 0;
    const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] += a[(i + 1) & 7] ^ 0x55u; // v_xor + v_add: 2 per element ... counted below
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = a[i] * a[(i + 1) & 7];
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] += b[(i + 1) & 3];
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) acc ^= prv_step(s0, l0, h0) ^ prv_step(s1, l1, h1);
        }
    }
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u64 v = acc ^ s0 ^ s1;
    for (int i = 0; i < 8; i++) v ^= a[i];
    for (int i = 0; i < 4; i++) v ^= b[i];
    out[tid] = v;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main(int argc, char **argv) {
    const int iters = 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    u64 *out, *stamps;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8 * 2);
    hipMalloc(&stamps, (size_t)cus * 8 * 2 * 8 * 2);
    // wave instructions per iteration and lane, read off the loops above
    const double per_iter[4] = {4 * 8 * 2, 4 * 8, 8 * 4 * 2, 8 * 2 /* steps */};
    const char *name[4] = {"v_xor_b32 + v_add_u32", "v_mul_lo_u32", "64-bit add (v_add_co + v_addc)", "prvhash steps"};
    for (int mode = 0; mode < 4; mode++)
        for (int waves : {1, 2, 4, 8}) {
            const int blocks = cus * waves; // 256 threads = 4 waves = one per SIMD; `waves` blocks per CU
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 1) k<1><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 2) k<2><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 3) k<3><<<blocks, 256>>>(out, stamps, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<u64> st(2 * blocks);
            hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
            double cyc = 0, ref = 0;
            for (int b = 0; b < blocks; b++) cyc += st[2 * b], ref += st[2 * b + 1];
            cyc /= blocks, ref /= blocks;
            const double units = per_iter[mode] * iters;            // per lane = per wave instruction stream
            const double per_simd_cycle = units * waves / cyc;      // `waves` waves share a SIMD
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, clock %.2f GHz, %.4f %s per SIMD cycle", mode, name[mode], waves, ms,
                   cyc / ref * 0.1, per_simd_cycle, mode == 3 ? "steps" : "wave instructions");
            if (mode == 3) printf(", %.3f T steps/s whole chip", units * 64.0 * 4 * blocks / (ms * 1e-3) / 1e12);
            printf("\n");
        }
    return 0;
}
