// Development probe (stand-alone: hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe).
// Measured on MI355X (profiles/r3_valu_rate_probe.txt): a 32-bit add / xor takes 2.1 SIMD cycles per wave instruction,
// v_mul_lo_u32 / v_mul_hi_u32 / v_lshl_add_u64 4.2, v_mad_u64_u32 4.5; the generator step as the compiler builds it 64
// cycles (2.42 T steps/s on the whole chip), on 32-bit halves with carry chains 81-89 (the carry instructions cost more
// than the 64-bit add they replace): the 64-bit form stays.
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

// the same step on 32-bit halves: the 64 x 64 -> 64 product as three low multiplies and one high multiply (no
// v_mad_u64_u32), the 64-bit additions as carry chains on the halves where they lie (the rotated seed is the two halves
// read in the other order: no register copies)
struct Prv32 {
    u32 sl, sh, ll, lh, hl, hh;
};
__device__ inline u64 prv_step32(Prv32 &g) {
    const u32 tl = (g.ll << 1) | 1u, th = __builtin_amdgcn_alignbit(g.lh, g.ll, 31);
    const u32 pl = g.sl * tl;
    const u32 ph = __umulhi(g.sl, tl) + g.sl * th + g.sh * tl;
    // rs = (lo = ph, hi = pl)
    u32 c;
    g.hl = __builtin_addc(g.hl, ph, 0u, &c);
    g.hh = __builtin_addc(g.hh, pl, c, &c);
    g.hl = __builtin_addc(g.hl, 0xAAAAAAAAu, 0u, &c);
    g.hh = __builtin_addc(g.hh, 0xAAAAAAAAu, c, &c);
    g.ll = __builtin_addc(g.ll, pl, 0u, &c);
    g.lh = __builtin_addc(g.lh, ph, c, &c);
    g.ll = __builtin_addc(g.ll, 0x55555555u, 0u, &c);
    g.lh = __builtin_addc(g.lh, 0x55555555u, c, &c);
    g.sl = pl ^ g.hl;
    g.sh = ph ^ g.hh;
    return ((u64)(g.lh ^ pl) << 32) | (g.ll ^ ph);
}

// ... and with the carry chains spelled out (the compiler turns some of the generic carry builtins into compare /
// select sequences)
__device__ inline u64 prv_step32_asm(Prv32 &g) {
    const u32 tl = (g.ll << 1) | 1u, th = __builtin_amdgcn_alignbit(g.lh, g.ll, 31);
    const u32 pl = g.sl * tl;
    const u32 ph = __umulhi(g.sl, tl) + g.sl * th + g.sh * tl;
    asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc\n\t"
        "v_add_co_u32 %0, vcc, %4, %0\n\tv_addc_co_u32 %1, vcc, %4, %1, vcc"
        : "+v"(g.hl), "+v"(g.hh)
        : "v"(ph), "v"(pl), "v"(0xAAAAAAAAu)
        : "vcc");
    asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc\n\t"
        "v_add_co_u32 %0, vcc, %4, %0\n\tv_addc_co_u32 %1, vcc, %4, %1, vcc"
        : "+v"(g.ll), "+v"(g.lh)
        : "v"(pl), "v"(ph), "v"(0x55555555u)
        : "vcc");
    g.sl = pl ^ g.hl;
    g.sh = ph ^ g.hh;
    return ((u64)(g.lh ^ pl) << 32) | (g.ll ^ ph);
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
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc ^= prv_step(s0, l0, h0) ^ prv_step(s1, l1, h1);
        } else if (MODE == 4) {
            Prv32 g0{(u32)s0, (u32)(s0 >> 32), (u32)l0, (u32)(l0 >> 32), (u32)h0, (u32)(h0 >> 32)};
            Prv32 g1{(u32)s1, (u32)(s1 >> 32), (u32)l1, (u32)(l1 >> 32), (u32)h1, (u32)(h1 >> 32)};
#pragma unroll
            for (int r = 0; r < 8; r++) acc ^= prv_step32(g0) ^ prv_step32(g1);
            s0 = ((u64)g0.sh << 32) | g0.sl, l0 = ((u64)g0.lh << 32) | g0.ll, h0 = ((u64)g0.hh << 32) | g0.hl;
            s1 = ((u64)g1.sh << 32) | g1.sl, l1 = ((u64)g1.lh << 32) | g1.ll, h1 = ((u64)g1.hh << 32) | g1.hl;
        } else if (MODE == 7) {
            Prv32 g0{(u32)s0, (u32)(s0 >> 32), (u32)l0, (u32)(l0 >> 32), (u32)h0, (u32)(h0 >> 32)};
            Prv32 g1{(u32)s1, (u32)(s1 >> 32), (u32)l1, (u32)(l1 >> 32), (u32)h1, (u32)(h1 >> 32)};
#pragma unroll
            for (int r = 0; r < 8; r++) acc ^= prv_step32_asm(g0) ^ prv_step32_asm(g1);
            s0 = ((u64)g0.sh << 32) | g0.sl, l0 = ((u64)g0.lh << 32) | g0.ll, h0 = ((u64)g0.hh << 32) | g0.hl;
            s1 = ((u64)g1.sh << 32) | g1.sl, l1 = ((u64)g1.lh << 32) | g1.ll, h1 = ((u64)g1.hh << 32) | g1.hl;
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = __umulhi(a[i], a[(i + 1) & 7]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = (u64)(u32)b[i] * (u64)(u32)b[(i + 1) & 3] + b[(i + 2) & 3]; // v_mad_u64_u32
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
    const double per_iter[8] = {4 * 8 * 2, 4 * 8, 8 * 4, 8 * 2 /* steps */, 8 * 2 /* steps */, 4 * 8, 8 * 4, 8 * 2};
    const char *name[8] = {"v_xor_b32 + v_add_u32", "v_mul_lo_u32", "64-bit add (v_lshl_add_u64)", "prvhash steps",
                           "prvhash steps on 32-bit halves", "v_mul_hi_u32", "v_mad_u64_u32",
                           "prvhash steps on 32-bit halves, carry chains in asm"};
    std::vector<u64> ref_out;
    for (int mode = 0; mode < 8; mode++)
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
                if (mode == 4) k<4><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 5) k<5><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 6) k<6><<<blocks, 256>>>(out, stamps, iters);
                if (mode == 7) k<7><<<blocks, 256>>>(out, stamps, iters);
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
            const double ghz = cyc / ref * 0.1;
            const double simd_cycles = ms * 1e-3 * ghz * 1e9;       // cycles every SIMD had during the launch
            const bool steps = mode == 3 || mode == 4 || mode == 7;
            if (steps && waves == 1) { // the three spellings of the step must give the same words
                std::vector<u64> o((size_t)blocks * 256);
                hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost);
                if (mode == 3) ref_out = o;
                else printf("mode %d results %s mode 3\n", mode, o == ref_out ? "EQUAL" : "DIFFER FROM");
            }
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, clock %.2f GHz, %.1f SIMD cycles per %s", mode, name[mode], waves, ms,
                   ghz, simd_cycles / (units * waves), steps ? "step of a wave" : "wave instruction");
            if (steps) printf(", %.3f T steps/s whole chip", units * 64.0 * 4 * blocks / (ms * 1e-3) / 1e12);
            printf("\n");
        }
    return 0;
}
