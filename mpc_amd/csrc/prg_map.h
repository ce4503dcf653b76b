// Consumers fused with the regeneration of PRgenerator streams (utils/PRgenerator.cpp:3-19): the stream words are
// handed to a functor as they are produced instead of being written to memory and read back.
//
// Geometry (the same as the fill kernels): a lane replays PRG_CK consecutive stream words from one checkpoint, so a
// wavefront owns 64 * PRG_CK consecutive words.  Every 16 steps the lanes park their words in LDS and the wavefront
// then visits those 64 x 16 words in memory order -- consecutive lanes get consecutive words -- so whatever the
// functor loads or stores at an index derived from the word's position is coalesced in 128-byte runs.  The exchange
// stays inside one wavefront (whose LDS operations complete in program order), so no workgroup barrier is needed and
// the waves of a workgroup drift apart: some generate while others store.
#pragma once
#include "mpcx_internal.h"

// orders the LDS traffic of one wavefront for the compiler; the hardware already runs it in program order
__device__ inline void prg_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

// A generator positioned anywhere in a stream: starts from the checkpoint below word t and discards the words in
// between.  Every lane of a launch opens a section at words that differ by multiples of PRG_CK, so the number of
// discarded words is the same for all of them.
struct PrgCursor {
    u64 s = 0, l = 0, h = 0;
    __device__ void open(const PrgCk *__restrict__ ck, u64 t, bool active) {
        if (active) {
            const PrgCk c = ck[t / PRG_CK];
            s = c.seed, l = c.lcg, h = c.hash;
        }
        for (u32 i = 0; i < (u32)(t % PRG_CK); i++) prv_step(s, l, h);
    }
    __device__ u64 next() { return prv_step(s, l, h); }
    // the same word, formed on the spot: left to itself the compiler may carry lcg and the rotated seed of every step of
    // an unrolled run along and xor them only where the word is used, four registers per step instead of two
    __device__ u64 take() {
        u64 w = prv_step(s, l, h);
        asm volatile("" : "+v"(w));
        return w;
    }
};

// PRG_CK / 2 bytes (uint16)int_buff % 67 from one checkpoint, packed four to a word
__device__ inline void mod67_run(const PrgCk c, int zero_to_one, u32 (&w)[PRG_CK / 8]) {
    u64 seed = c.seed, lcg = c.lcg, hash = c.hash;
#pragma unroll
    for (u32 q = 0; q < PRG_CK / 8; q++) {
        u32 acc = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 x = (u32)prv_step(seed, lcg, hash) & 0xFFFFu;
            (void)prv_step(seed, lcg, hash);
            // x % 67 for x < 2^16: floor(x / 67) = (x * 62602) >> 22 exactly, in 24-bit multiplies
            u32 m = x - (u32)__umul24((u32)__umul24(x, 62602u) >> 22, 67u);
            if (zero_to_one && m == 0) m = 1;
            acc |= m << (8 * k);
        }
        w[q] = acc;
    }
}

// f(rel, v) for rel in [0, count) with v = g(A[tA + rel], B[tB + rel]) (TWO streams) or v = A[tA + rel] (one).
// g runs in the producing lane (registers only); f runs in memory order.  A lane starts from the checkpoint below
// its first word and discards the words in between (the same number for every lane).
template <bool TWO, class G, class F>
__global__ __launch_bounds__(256) void prg_map_kernel(const PrgCk *__restrict__ ckA, const PrgCk *__restrict__ ckB,
                                                      u64 tA, u64 tB, u64 count, G g, F f) {
    constexpr u32 CH = 16; // steps between two visits (8: 64-byte runs, 32: half the waves per CU; both measured slower)
    __shared__ u64 tile[256 * (CH + 1)];
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
    for (u32 it = 0; it < PRG_CK / CH; it++) {
#pragma unroll
        for (u32 k = 0; k < CH; k++) {
            const u64 wa = prv_step(sa, la, ha);
            tile[tid * (CH + 1) + k] = TWO ? g(wa, prv_step(sb, lb, hb)) : wa;
        }
        prg_wave_sync();
#pragma unroll 4
        for (u32 j = 0; j < CH; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / CH, col = idx % CH;
            const u64 rel = (blk_r0 + row) * PRG_CK + it * CH + col;
            if (rel < count) f(rel, tile[row * (CH + 1) + col]);
        }
        prg_wave_sync();
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

// ---- buffer form: int_buff[i] is stream word 2i and bool_buff[i] the low bit of word 2i + 1 of a generator
// (utils/PRgenerator.cpp:27-32, extended to every i).  Element i of the call reads int_buff[first[s] + i] of up to three
// sections s of ONE generator (and bool_buff[first[0] + i]); g folds them, in the producing lane, into a value V
// (trivially copyable, a multiple of 8 bytes) and f(i, V) runs in memory order.  A lane owns PRG_CK / 2 consecutive
// elements; every 16 (or 8) elements the wavefront visits its 64 x 16 values with consecutive lanes on consecutive elements.
// Elements per visit: 16 (128-byte runs of 8-byte words) while the tile stays near 64 KiB, else 8; CHV overrides it
// (measured per consumer: a light consumer of a two-word value prefers the smaller tile and twice the workgroups).
template <int NS, class V, u32 CHV, class G, class F>
__global__ __launch_bounds__(256) void prg_buff_map_kernel(const PrgCk *__restrict__ ck, u64 f0, u64 f1, u64 f2, u64 count,
                                                           G g, F f) {
    static_assert(sizeof(V) % 8 == 0 && NS >= 1 && NS <= 3, "value layout");
    constexpr u32 VW = sizeof(V) / 8, EPL = PRG_CK / 2, CH = CHV ? CHV : (VW <= 2 ? 16 : 8), PITCH = CH * VW + 1;
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 blk_r0 = (u64)blockIdx.x * 256;
    const u64 run = blk_r0 + tid;
    const bool active = run * EPL < count;
    const u64 first[3] = {f0, f1, f2};
    u64 sd[NS], lc[NS], hs[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        sd[s] = lc[s] = hs[s] = 0;
        const u64 t = 2 * (first[s] + run * EPL);
        if (active) {
            const PrgCk c = ck[t / PRG_CK];
            sd[s] = c.seed, lc[s] = c.lcg, hs[s] = c.hash;
        }
        const u32 skip = (u32)((2 * first[s]) % PRG_CK); // the same for every lane
        for (u32 i = 0; i < skip; i++) prv_step(sd[s], lc[s], hs[s]);
    }
    for (u32 it = 0; it < EPL / CH; it++) {
#pragma unroll
        for (u32 e = 0; e < CH; e++) {
            u64 w[NS];
            u32 b0 = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                w[s] = prv_step(sd[s], lc[s], hs[s]);
                const u64 odd = prv_step(sd[s], lc[s], hs[s]);
                if (s == 0) b0 = (u32)(odd & 1);
            }
            *(V *)&tile[tid * PITCH + e * VW] = g(w, b0);
        }
        prg_wave_sync();
#pragma unroll 2
        for (u32 j = 0; j < CH; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / CH, col = idx % CH;
            const u64 i = (blk_r0 + row) * EPL + it * CH + col;
            if (i < count) f(i, *(const V *)&tile[row * PITCH + col * VW]);
        }
        prg_wave_sync();
    }
}

template <int NS, class V, u32 CHV = 0, class G, class F>
int prg_buff_map(mpcx_ctx *ctx, u64 seed, const u64 (&first)[NS], u64 count, G g, F f) {
    if (count == 0) return 0;
    u64 top = 0;
    for (int s = 0; s < NS; s++) top = first[s] > top ? first[s] : top;
    const PrgCk *ck;
    MPCX_TRY(mpcx_prg_table(ctx, seed, 0, 2 * (top + count) + PRG_CK, &ck));
    const u64 grid = ceil_div(ceil_div(count, PRG_CK / 2), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "prg map: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL((prg_buff_map_kernel<NS, V, CHV, G, F>), dim3((u32)grid), dim3(256), 0, ctx->stream, ck, first[0],
                       NS > 1 ? first[NS > 1 ? 1 : 0] : 0, NS > 2 ? first[NS > 2 ? 2 : 0] : 0, count, g, f);
    MPCX_LAUNCH_CHECK();
    return 0;
}
