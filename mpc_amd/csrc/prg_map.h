// Consumers fused with the regeneration of PRgenerator streams (utils/PRgenerator.cpp:3-19): the stream words are
// handed to a functor as they are produced instead of being written to memory and read back.
//
// Geometry (the same as the fill kernels): a lane replays PRG_CK consecutive stream words from one checkpoint, so a
// 256-thread workgroup owns 256 * PRG_CK consecutive words.  Every 16 steps the lanes park their words in LDS and
// the workgroup then visits those 256 x 16 words in memory order -- consecutive threads get consecutive words -- so
// whatever the functor loads or stores at an index derived from the word's position is coalesced in 128-byte runs.
#pragma once
#include "mpcx_internal.h"

__host__ __device__ inline u64 prv_step(u64 &seed, u64 &lcg, u64 &hash) {
    seed *= lcg * 2 + 1;
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the product in registers as one value: otherwise the compiler re-derives it inside `lcg += seed`
    // with a second 64-bit multiply-add, which costs more than the plain 64-bit add it replaces
    asm volatile("" : "+v"(seed));
#endif
    const u64 rs = seed >> 32 | seed << 32;
    hash += rs + 0xAAAAAAAAAAAAAAAAull;
    lcg += seed + 0x5555555555555555ull;
    seed ^= hash;
    return lcg ^ rs;
}

// f(rel, v) for rel in [0, count) with v = g(A[tA + rel], B[tB + rel]) (TWO streams) or v = A[tA + rel] (one).
// g runs in the producing lane (registers only); f runs in memory order.  A lane starts from the checkpoint below
// its first word and discards the words in between (the same number for every lane).
template <bool TWO, class G, class F>
__global__ __launch_bounds__(256) void prg_map_kernel(const PrgCk *__restrict__ ckA, const PrgCk *__restrict__ ckB,
                                                      u64 tA, u64 tB, u64 count, G g, F f) {
    __shared__ u64 tile[256 * 17];
    const u32 tid = threadIdx.x;
    const u64 blk_r0 = (u64)blockIdx.x * 256; // first run of PRG_CK words of this workgroup
    const u64 run = blk_r0 + tid;
    const bool active = run * PRG_CK < count;
    u64 sa = 0, la = 0, ha = 0, sb = 0, lb = 0, hb = 0;
    const u32 skipA = (u32)(tA % PRG_CK), skipB = TWO ? (u32)(tB % PRG_CK) : 0;
    if (active) {
        const PrgCk a = ckA[(tA + run * PRG_CK) / PRG_CK];
        sa = a.seed, la = a.lcg, ha = a.hash;
        if (TWO) {
            const PrgCk b = ckB[(tB + run * PRG_CK) / PRG_CK];
            sb = b.seed, lb = b.lcg, hb = b.hash;
        }
    }
    for (u32 i = 0; i < skipA; i++) prv_step(sa, la, ha);
    if (TWO)
        for (u32 i = 0; i < skipB; i++) prv_step(sb, lb, hb);
    for (u32 it = 0; it < PRG_CK / 16; it++) {
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 wa = prv_step(sa, la, ha);
            tile[tid * 17 + k] = TWO ? g(wa, prv_step(sb, lb, hb)) : wa;
        }
        __syncthreads();
#pragma unroll 4
        for (u32 j = 0; j < 16; j++) {
            const u32 idx = j * 256 + tid;
            const u32 row = idx >> 4, col = idx & 15;
            const u64 rel = (blk_r0 + row) * PRG_CK + it * 16 + col;
            if (rel < count) f(rel, tile[row * 17 + col]);
        }
        __syncthreads();
    }
}

struct PrgNoCombine {
    __device__ u64 operator()(u64 a, u64) const { return a; }
};

// f(rel, word) over stream words [t_first, t_first + count) of generator (seed, 0); asynchronous on ctx->stream
template <class F> int prg_map1(mpcx_ctx *ctx, u64 seed, u64 t_first, u64 count, F f) {
    if (count == 0) return 0;
    const PrgCk *ck;
    MPCX_TRY(mpcx_prg_table(ctx, seed, 0, t_first + count, &ck));
    const u64 grid = ceil_div(ceil_div(count, PRG_CK), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "prg map: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL((prg_map_kernel<false, PrgNoCombine, F>), dim3((u32)grid), dim3(256), 0, ctx->stream, ck, ck, t_first,
                       t_first, count, PrgNoCombine{}, f);
    MPCX_LAUNCH_CHECK();
    return 0;
}
// f(rel, g(A word, B word)) over words [tA, tA + count) of generator (seedA, 0) and [tB, tB + count) of (seedB, 0)
template <class G, class F>
int prg_map2(mpcx_ctx *ctx, u64 seedA, u64 tA, u64 seedB, u64 tB, u64 count, G g, F f) {
    if (count == 0) return 0;
    const PrgCk *ckA, *ckB;
    MPCX_TRY(mpcx_prg_table(ctx, seedA, 0, tA + count, &ckA));
    MPCX_TRY(mpcx_prg_table(ctx, seedB, 0, tB + count, &ckB));
    const u64 grid = ceil_div(ceil_div(count, PRG_CK), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "prg map: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL((prg_map_kernel<true, G, F>), dim3((u32)grid), dim3(256), 0, ctx->stream, ckA, ckB, tA, tB, count, g, f);
    MPCX_LAUNCH_CHECK();
    return 0;
}
