// Per-party local segments of ABY3Engine (mpc_engine/ABY3Engine.cpp) as HIP kernels.
// All kernels are streaming, HBM-bound passes over 64-bit share words; the carry tree of msb() is
// kept as dense bit strings that halve every round, only the wire messages are expanded to one byte per
// bit, and the pop-stream words a segment needs are consumed while they are regenerated (prg_map.h).
#include "mpcx_internal.h"
#include "prg_map.h"

namespace {

template <class F> __global__ __launch_bounds__(256) void ew_kernel(u64 n, F f) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) f(i);
}
template <class F> int ew_launch(mpcx_ctx *ctx, u64 n, F f) {
    if (n == 0) return 0;
    u64 blocks = ceil_div(n, 256);
    if (blocks > MPCX_MAX_GRID) blocks = MPCX_MAX_GRID; // one pass per lane measures faster than a capped, grid-strided launch
    hipLaunchKernelGGL(ew_kernel<F>, dim3((u32)blocks), dim3(256), 0, ctx->stream, n, f);
    MPCX_LAUNCH_CHECK();
    return 0;
}

// ---- msb() carry tree (ABY3Engine.cpp:361-467) on dense bit arrays ------------------------------------------
// The reference keeps one byte per bit and halves the vectors every round.  Here every share array (G0, G1, P0,
// P1) of a level is ONE dense bit string: before the round with `temp` pairs per element, element k owns bits
// [2*temp*k, 2*temp*(k+1)).  Pairs never straddle elements (2*temp is even), so a round is the same operation on
// every aligned bit pair of the string, whatever the level: a lane takes 32 bits of each array, forms the 32
// wire bytes they stand for (wire byte b of the whole message = bit b of the result string) and writes the 16
// bits they leave in the next level.  `state` is 48*n bytes; levels alternate between two regions so that a round
// never writes what another lane of the same launch still reads:
//   level (pairs)    : 32   16    8    4    2    1    0 (result, 1 bit per element)
//   region at byte   : --   32n   0    32n  0    32n  0
// and the four arrays of a level sit msb_stride() bytes apart.  Level 32 is never stored: its rounds read the
// caller's G / P words in place (their addresses are kept in the first 32 bytes of `state`) and apply
// ABY3Engine.cpp:385-387 (G[63] = 0, P[63] = 1) on the fly.
inline u64 msb_stride(u64 n, u32 temp) {
    const u64 bits = temp ? 2ull * temp * n : n;
    return (ceil_div(bits, 8) + 3) & ~3ull; // whole 32-bit words
}
inline u64 msb_base(u64 n, u32 temp) { return (temp == 32 || temp == 8 || temp == 2 || temp == 0) ? 0 : 32 * n; }
// the backup bits of P0 / P1 (bit 63 of every element, :385): two arrays of one byte per element behind the levels of
// the lower region (the largest of them, level 8, ends at or before byte 16 n)
inline u64 msb_backup_base(u64 n) { return 16 * n; }
inline u64 msb_backup_stride(u64 n) { return (n + 7) & ~7ull; }

// bits 0,2,4,.. of x moved to bits 0,1,2,..
__device__ inline u32 compress_even(u32 x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// 4 bits -> 4 bytes of 0/1 (the shifted copies of a 4-bit value do not overlap, so the product has no carries)
__device__ inline u32 nibble_to_bytes(u32 x) { return (u32)__umul24(x & 0xFu, 0x204081u) & 0x01010101u; }
// 4 bytes (bit 0 of each counts) -> 4 bits
__device__ inline u32 bytes_to_nibble(u32 w) {
    w &= 0x01010101u;
    w |= w >> 7;
    w |= w >> 14;
    return w & 0xFu;
}

// Before the round (:397-414 + :253-256).  For the pair (2m, 2m+1): result bit 2m = share of P[2m+1] & G[2m],
// result bit 2m+1 = share of P[2m+1] & P[2m], shared AND = (x0&y0)^(x1&y0)^(x0&y1) without re-randomisation (:255).
// The result string goes out as one byte per bit; share 0 of the next level (:424-425 with i = 0) is
// G' = result_even ^ G_odd, P' = result_odd, compacted.
//
// The four share arrays a round reads: G0, G1, P0, P1 as 32-bit words.  FIRST (level 32): the caller's own G / P
// words, with bit 63 of every element (bit 31 of the odd words) forced to 0 in G and 1 in P.
struct MsbIn {
    const u32 *a[4];
};
template <bool FIRST> __device__ inline u32 msb_in_word(const MsbIn &in, u32 arr, u64 i) {
    const u32 v = in.a[arr][i];
    if (!FIRST || !(i & 1)) return v;
    return arr < 2 ? v & 0x7FFFFFFFu : v | 0x80000000u;
}
// 32 bits of a result string -> its 32 wire bytes
template <bool ALIGNED16> __device__ inline void msb_put32(u8 *dst, u32 r) {
    u32 w[8];
#pragma unroll
    for (u32 q = 0; q < 8; q++) w[q] = nibble_to_bytes(r >> (4 * q));
    if (ALIGNED16) {
        *(uint4 *)dst = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
#pragma unroll
        for (u32 q = 0; q < 4; q++) *(uint2 *)(dst + 8 * q) = make_uint2(w[2 * q], w[2 * q + 1]);
    }
}
// 32 wire bytes -> 32 bits
template <bool ALIGNED16> __device__ inline u32 msb_get32(const u8 *src) {
    u32 w[8];
    if (ALIGNED16) {
        const uint4 a = *(const uint4 *)src, b = *(const uint4 *)(src + 16);
        w[0] = a.x, w[1] = a.y, w[2] = a.z, w[3] = a.w, w[4] = b.x, w[5] = b.y, w[6] = b.z, w[7] = b.w;
    } else {
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            const uint2 a = *(const uint2 *)(src + 8 * q);
            w[2 * q] = a.x, w[2 * q + 1] = a.y;
        }
    }
    u32 r = 0;
#pragma unroll
    for (u32 q = 0; q < 8; q++) r |= bytes_to_nibble(w[q]) << (4 * q);
    return r;
}
// the product share of one round on 32 bits of the four arrays: even result bits in re, odd ones (unshifted) in ro
__device__ inline void msb_and32(u32 g0, u32 g1, u32 p0, u32 p1, u32 &re, u32 &ro) {
    constexpr u32 E = 0x55555555u;
    const u32 po0 = (p0 >> 1) & E, po1 = (p1 >> 1) & E;
    const u32 ge0 = g0 & E, ge1 = g1 & E, pe0 = p0 & E, pe1 = p1 & E;
    re = (po0 & ge0) ^ (po1 & ge0) ^ (po0 & ge1);
    ro = (po0 & pe0) ^ (po1 & pe0) ^ (po0 & pe1);
}

// FIRST also keeps bit 63 of every P word (`backup`, :385) in two byte arrays bk0 / bk1 for the output step.
// PACKED: the message stays a bit string (bit b = wire byte b), 32 bits per lane as one word; the bytes of the wire
// are made from it on the host (mpcx_d2h_bits), so an eighth of the bytes cross the link.
template <bool ALIGNED16, bool FIRST, bool PACKED = false>
__global__ __launch_bounds__(256) void msb_round_pre_kernel(MsbIn in, u8 *__restrict__ out, u64 sout, u64 nbits,
                                                            u8 *__restrict__ msg, u8 *__restrict__ bk0,
                                                            u8 *__restrict__ bk1) {
    constexpr u32 E = 0x55555555u;
    const u64 words = (nbits + 31) / 32;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) {
        const u32 g0 = msb_in_word<FIRST>(in, 0, i), g1 = msb_in_word<FIRST>(in, 1, i);
        const u32 p0 = msb_in_word<FIRST>(in, 2, i), p1 = msb_in_word<FIRST>(in, 3, i);
        if (FIRST && (i & 1)) { // the raw top bits (msb_in_word has forced them to 1)
            bk0[i >> 1] = (u8)(in.a[2][i] >> 31);
            bk1[i >> 1] = (u8)(in.a[3][i] >> 31);
        }
        u32 re, ro;
        msb_and32(g0, g1, p0, p1, re, ro);
        const u32 r = re | (ro << 1);
        if (PACKED) {
            const u64 left = nbits - 32 * i; // bits of the message in this word
            ((u32 *)msg)[i] = left >= 32 ? r : r & ((1u << left) - 1u);
        } else if (32 * i + 32 <= nbits) {
            msb_put32<ALIGNED16>(msg + 32 * i, r);
        } else {
            for (u64 t = 32 * i; t < nbits; t++) msg[t] = (u8)((r >> (t & 31)) & 1u);
        }
        if (2 * i + 2 <= sout) {
            *(u16 *)(out + 2 * i) = (u16)compress_even(re ^ ((g0 >> 1) & E));
            *(u16 *)(out + 2 * sout + 2 * i) = (u16)compress_even(ro);
        }
    }
}
// After the round (:416-428 with i = 1): msg = result[1] from the next party, packed back into bits; share 1 of the
// next level is written.
template <bool ALIGNED16, bool FIRST>
__global__ __launch_bounds__(256) void msb_round_post_kernel(MsbIn in, u8 *__restrict__ out, u64 sout, u64 nbits,
                                                             const u8 *__restrict__ msg) {
    constexpr u32 E = 0x55555555u;
    const u64 words = (nbits + 31) / 32;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) {
        u32 r = 0;
        if (32 * i + 32 <= nbits) {
            r = msb_get32<ALIGNED16>(msg + 32 * i);
        } else {
            for (u64 t = 32 * i; t < nbits; t++) r |= (u32)(msg[t] & 1) << (t & 31);
        }
        const u32 g1 = msb_in_word<FIRST>(in, 1, i);
        if (2 * i + 2 <= sout) {
            *(u16 *)(out + sout + 2 * i) = (u16)compress_even((r & E) ^ ((g1 >> 1) & E));
            *(u16 *)(out + 3 * sout + 2 * i) = (u16)compress_even((r >> 1) & E);
        }
    }
}
// The whole local segment between two messenger rounds of the tree: what round_post(T) and round_pre(T / 2) do, in
// one pass and without storing the level in between.  A lane takes 64 bits of the level-T strings: the 64 wire
// bytes of result[1] and the 64 bits of G1 give 32 bits of share 1 of level T/2 (:416-428), which stay in registers;
// with the 32 bits of share 0 that the previous segment stored they give this party's product share of round T/2
// (:397-414, :253-256) -- 32 wire bytes -- and 16 bits of share 0 of level T/4.  Of share 1 only G is stored (the
// next segment reads its odd bits); P1 of a level is never needed again.
template <bool ALIGNED16, bool FIRST, bool PACKED = false>
__global__ __launch_bounds__(256) void msb_round_mid_kernel(const u32 *__restrict__ g1_in, u64 g1_words,
                                                            u8 *__restrict__ half, u64 shalf, u8 *__restrict__ quarter,
                                                            u64 squarter, u64 nbits, const u8 *__restrict__ msg_in,
                                                            u8 *__restrict__ msg_out) {
    constexpr u32 E = 0x55555555u;
    const u64 lanes = (nbits + 63) / 64;
    const u64 hbits = nbits / 2; // bits of a level-T/2 string = bytes of the outgoing message
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < lanes; i += (u64)gridDim.x * 256) {
        u32 rlo = 0, rhi = 0;
        if (PACKED) { // bits beyond the message are zero (mpcx_h2d_bits)
            rlo = ((const u32 *)msg_in)[2 * i];
            rhi = 64 * i + 32 < nbits ? ((const u32 *)msg_in)[2 * i + 1] : 0;
        } else if (64 * i + 64 <= nbits) {
            rlo = msb_get32<ALIGNED16>(msg_in + 64 * i);
            rhi = msb_get32<ALIGNED16>(msg_in + 64 * i + 32);
        } else {
            for (u64 t = 64 * i; t < nbits; t++) {
                const u32 b = (u32)(msg_in[t] & 1);
                if ((t & 63) < 32)
                    rlo |= b << (t & 31);
                else
                    rhi |= b << (t & 31);
            }
        }
        u32 glo = g1_in[2 * i], ghi = 2 * i + 1 < g1_words ? g1_in[2 * i + 1] : 0;
        if (FIRST) ghi &= 0x7FFFFFFFu;
        // share 1 of level T/2
        const u32 G1h = compress_even((rlo & E) ^ ((glo >> 1) & E)) | compress_even((rhi & E) ^ ((ghi >> 1) & E)) << 16;
        const u32 P1h = compress_even((rlo >> 1) & E) | compress_even((rhi >> 1) & E) << 16;
        ((u32 *)(half + shalf))[i] = G1h;
        const u32 G0h = ((const u32 *)half)[i], P0h = ((const u32 *)(half + 2 * shalf))[i];
        u32 re, ro;
        msb_and32(G0h, G1h, P0h, P1h, re, ro);
        const u32 r = re | (ro << 1);
        if (PACKED) {
            const u64 left = hbits - 32 * i;
            ((u32 *)msg_out)[i] = left >= 32 ? r : r & ((1u << left) - 1u);
        } else if (32 * i + 32 <= hbits) {
            msb_put32<ALIGNED16>(msg_out + 32 * i, r);
        } else {
            for (u64 t = 32 * i; t < hbits; t++) msg_out[t] = (u8)((r >> (t & 31)) & 1u);
        }
        if (2 * i + 2 <= squarter) {
            *(u16 *)(quarter + 2 * i) = (u16)compress_even(re ^ ((G0h >> 1) & E));
            *(u16 *)(quarter + 2 * squarter + 2 * i) = (u16)compress_even(ro);
        }
    }
}
// The segment after the last round: round_post(1) and the output step (:459-466) in one pass, eight elements per
// lane.  msg = result[1] of round 1 (two bytes per element); g1_lvl1 = G1 of level 1 (two bits per element),
// g0_lvl0 = G0 of level 0 (one bit per element); bk0 / bk1 = the backup bytes the first round kept.
template <bool ALIGNED, bool PACKED = false>
__global__ __launch_bounds__(256) void msb_last_kernel(const u8 *__restrict__ g1_lvl1, const u8 *__restrict__ g0_lvl0,
                                                       const u8 *__restrict__ msg, const u8 *__restrict__ bk0,
                                                       const u8 *__restrict__ bk1, u64 n, u8 *__restrict__ Z0,
                                                       u8 *__restrict__ Z1) {
    const u64 groups = (n + 7) / 8;
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < groups; q += (u64)gridDim.x * 256) {
        const u64 k0 = 8 * q;
        const u32 g1w = g1_lvl1[2 * q] | (k0 + 4 < n ? (u32)g1_lvl1[2 * q + 1] << 8 : 0), g0w = g0_lvl0[q];
        if (PACKED) { // two message bits per element: the lane's eight elements are one 16-bit word
            const u32 mw16 = ((const u16 *)msg)[q];
            for (u64 k = k0; k < n && k < k0 + 8; k++) {
                const u32 e = (u32)(k - k0);
                Z0[k] = (u8)(((g0w >> e) & 1u) ^ bk0[k]);
                Z1[k] = (u8)((((mw16 >> (2 * e)) & 1u) ^ ((g1w >> (2 * e + 1)) & 1u)) ^ bk1[k]);
            }
        } else if (ALIGNED && k0 + 8 <= n) {
            const uint4 m = *(const uint4 *)(msg + 2 * k0);
            const u32 mw[4] = {m.x, m.y, m.z, m.w};
            u64 z0 = *(const u64 *)(bk0 + k0), z1 = *(const u64 *)(bk1 + k0);
#pragma unroll
            for (u32 e = 0; e < 8; e++) {
                const u64 g1 = ((mw[e >> 1] >> (16 * (e & 1))) & 1u) ^ ((g1w >> (2 * e + 1)) & 1u); // :424 with i = 1
                z0 ^= (u64)((g0w >> e) & 1u) << (8 * e);
                z1 ^= g1 << (8 * e);
            }
            *(u64 *)(Z0 + k0) = z0;
            *(u64 *)(Z1 + k0) = z1;
        } else {
            for (u64 k = k0; k < n && k < k0 + 8; k++) {
                const u32 e = (u32)(k - k0);
                Z0[k] = (u8)(((g0w >> e) & 1u) ^ bk0[k]);
                Z1[k] = (u8)(((u32)(msg[2 * k] & 1) ^ ((g1w >> (2 * e + 1)) & 1u)) ^ bk1[k]);
            }
        }
    }
}

inline u32 grid_for(u64 work) { // one item per lane unless the launch limit forces the kernels to loop
    u64 b = ceil_div(work, 256);
    if (b > MPCX_MAX_GRID) b = MPCX_MAX_GRID;
    if (b == 0) b = 1;
    return (u32)b;
}

} // namespace

extern "C" {

int mpcx_ring_add(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && out) || n == 0, "mpcx_ring_add: NULL pointer");
    const u64 *x = (const u64 *)a, *y = (const u64 *)b;
    u64 *o = (u64 *)out;
    return ew_launch(ctx, n, [=] __device__(u64 i) { o[i] = x[i] + y[i]; });
}
int mpcx_ring_sub(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && out) || n == 0, "mpcx_ring_sub: NULL pointer");
    const u64 *x = (const u64 *)a, *y = (const u64 *)b;
    u64 *o = (u64 *)out;
    return ew_launch(ctx, n, [=] __device__(u64 i) { o[i] = x[i] - y[i]; });
}
int mpcx_ring_sar(mpcx_ctx *ctx, const int64_t *a, int64_t *out, uint64_t n, int k) {
    MPCX_REQUIRE((a && out) || n == 0, "mpcx_ring_sar: NULL pointer");
    MPCX_REQUIRE(k >= 0 && k < 64, "mpcx_ring_sar: shift %d out of range", k);
    return ew_launch(ctx, n, [=] __device__(u64 i) { out[i] = a[i] >> k; });
}
int mpcx_ring_shl1(mpcx_ctx *ctx, const int64_t *a, int64_t *out, uint64_t n) {
    MPCX_REQUIRE((a && out) || n == 0, "mpcx_ring_shl1: NULL pointer");
    const u64 *x = (const u64 *)a;
    u64 *o = (u64 *)out;
    return ew_launch(ctx, n, [=] __device__(u64 i) { o[i] = x[i] << 1; });
}

int mpcx_aby3_share_local(mpcx_ctx *ctx, const int64_t *X, uint64_t n, uint64_t seed_next, uint64_t pop_next,
                          uint64_t seed_prev, uint64_t pop_prev, int64_t *out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(out, "mpcx_aby3_share_local: out is NULL");
    // the two pop streams are regenerated, subtracted and added to X in one pass
    return prg_fill_diff(ctx, seed_next, PRG_POP_T0 + pop_next, seed_prev, PRG_POP_T0 + pop_prev, n, X, out);
}

int mpcx_aby3_share_bools_local(mpcx_ctx *ctx, const int8_t *X, uint64_t n, uint64_t seed_next, uint64_t pop_next,
                                uint64_t seed_prev, uint64_t pop_prev, int8_t *out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(out, "mpcx_aby3_share_bools_local: out is NULL");
    auto bit = [] __device__(u64 a, u64 b) { return (a ^ b) & 1; };
    const u64 tn = PRG_POP_T0 + pop_next, tp = PRG_POP_T0 + pop_prev;
    if (X) return prg_map2(ctx, seed_next, tn, seed_prev, tp, n, bit, [=] __device__(u64 i, u64 r) { out[i] = (i8)((i8)r ^ X[i]); });
    return prg_map2(ctx, seed_next, tn, seed_prev, tp, n, bit, [=] __device__(u64 i, u64 r) { out[i] = (i8)r; });
}

// Z0 = X0*(Y0+Y1) + X1*Y0 = [X0 | X1] * [Y0+Y1 ; Y0]  (ABY3Engine.cpp:189) as ONE K-concatenated ring
// product, with the mask pop(next) - pop(prev) (:190-197) added in the product's epilogue.
static int aby3_matmul_impl(mpcx_ctx *ctx, const i64 *X0, const i64 *X1, const i64 *Y0, const i64 *Y1, u64 rows, u64 l,
                            u64 c, int layoutX, int layoutY, u64 first_word, u64 seed_next, u64 pop_next, u64 seed_prev,
                            u64 pop_prev, i64 *Z0, const RnsPreB *pre = nullptr) {
    // the mask words go straight into Z0 and the product is accumulated onto them
    MPCX_TRY(prg_fill_diff(ctx, seed_next, PRG_POP_T0 + pop_next + first_word, seed_prev, PRG_POP_T0 + pop_prev + first_word,
                           rows * c, nullptr, Z0));
    GemmPart parts[2] = {{X0, Y0, Y1}, {X1, Y0, nullptr}};
    return ring_gemm_parts(ctx, parts, 2, rows, l, c, layoutX, layoutY, nullptr, nullptr, Z0, 0, 1, pre);
}

int mpcx_aby3_matmul_local_rows(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0,
                                const int64_t *Y1, uint64_t rows, uint64_t l, uint64_t c, uint64_t row0,
                                uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                                int64_t *Z0) {
    MPCX_REQUIRE(X0 && X1 && Y0 && Y1 && Z0, "mpcx_aby3_matmul_local_rows: NULL operand");
    return aby3_matmul_impl(ctx, X0, X1, Y0, Y1, rows, l, c, 0, 0, row0 * c, seed_next, pop_next, seed_prev, pop_prev, Z0);
}

int mpcx_aby3_matmul_local(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                           uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next, uint64_t pop_next,
                           uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0) {
    MPCX_REQUIRE(X0 && X1 && Y0 && Y1 && Z0, "mpcx_aby3_matmul_local: NULL operand");
    MPCX_REQUIRE(layout == 0 || layout == 1, "mpcx_aby3_matmul_local: layout must be 0 or 1");
    return aby3_matmul_impl(ctx, X0, X1, Y0, Y1, r, l, c, layout, layout, 0, seed_next, pop_next, seed_prev, pop_prev, Z0);
}

// A second operand that many products use (weights): the share pair stays where the caller keeps it on the device, and
// when products of this shape can run in the residue form, the residue planes of [Y0 + Y1 ; Y0] are made once here.
struct mpcx_aby3_operand {
    const i64 *Y0, *Y1;
    u64 l, c;
    int layout;
    int device; // the planes live on the device of the context that made them
    RnsPreB pre;
};
int mpcx_aby3_operand_prepare(mpcx_ctx *ctx, const int64_t *Y0, const int64_t *Y1, uint64_t l, uint64_t c, int layout,
                              mpcx_aby3_operand **out) {
    MPCX_REQUIRE(ctx && Y0 && Y1 && out, "mpcx_aby3_operand_prepare: NULL argument");
    MPCX_REQUIRE(l && c, "mpcx_aby3_operand_prepare: empty shape");
    MPCX_REQUIRE(layout == 0 || layout == 1, "mpcx_aby3_operand_prepare: layout must be 0 or 1");
    mpcx_aby3_operand *op = new mpcx_aby3_operand{Y0, Y1, l, c, layout, ctx->device, RnsPreB()};
    // worth the memory (18 bytes per word of [Y0 + Y1 ; Y0] against 16 for the pair) only where a product with many rows
    // would take the residue form at all
    if (ring_gemm_use_rns(2, 16384, l, c)) {
        const GemmPart parts[2] = {{nullptr, Y0, Y1}, {nullptr, Y0, nullptr}};
        if (ring_gemm_rns_prepare_b(ctx, parts, 2, l, c, layout, &op->pre) != 0) {
            delete op;
            return 1;
        }
    }
    *out = op;
    return 0;
}
int mpcx_aby3_operand_free(mpcx_ctx *ctx, mpcx_aby3_operand *op) {
    if (!op) return 0;
    MPCX_REQUIRE(ctx, "mpcx_aby3_operand_free: the context is NULL");
    ring_gemm_rns_release_b(ctx, &op->pre);
    delete op;
    return 0;
}
uint64_t mpcx_aby3_operand_bytes(const mpcx_aby3_operand *op) { return op ? op->pre.nchunks * op->pre.chunk_bytes : 0; }
uint64_t mpcx_aby3_operand_uses(const mpcx_aby3_operand *op) { return op ? op->pre.used : 0; }
int mpcx_aby3_matmul_local_prepared(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, int layoutX,
                                    const mpcx_aby3_operand *Y, uint64_t rows, uint64_t row0, uint64_t seed_next,
                                    uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0) {
    MPCX_REQUIRE(X0 && X1 && Y && Z0, "mpcx_aby3_matmul_local_prepared: NULL operand");
    MPCX_REQUIRE(layoutX == 0 || layoutX == 1, "mpcx_aby3_matmul_local_prepared: layout must be 0 or 1");
    MPCX_REQUIRE(Y->device == ctx->device, "mpcx_aby3_matmul_local_prepared: the operand was prepared on device %d, the context "
                 "works on device %d", Y->device, ctx->device);
    return aby3_matmul_impl(ctx, X0, X1, Y->Y0, Y->Y1, rows, Y->l, Y->c, layoutX, Y->layout, row0 * Y->c, seed_next, pop_next,
                            seed_prev, pop_prev, Z0, &Y->pre);
}

int mpcx_aby3_vecmul_local(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                           uint64_t n, uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                           int64_t *Z0) {
    if (n == 0) return 0;
    MPCX_REQUIRE(X0 && X1 && Y0 && Y1 && Z0, "mpcx_aby3_vecmul_local: NULL operand");
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, 8 * n, &ws));
    u64 *m = (u64 *)ws;
    MPCX_TRY(prg_fill_diff(ctx, seed_next, PRG_POP_T0 + pop_next, seed_prev, PRG_POP_T0 + pop_prev, n, nullptr, (i64 *)m));
    const u64 *x0 = (const u64 *)X0, *x1 = (const u64 *)X1, *y0 = (const u64 *)Y0, *y1 = (const u64 *)Y1;
    u64 *z = (u64 *)Z0;
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        z[i] = x0[i] * y0[i] + x0[i] * y1[i] + x1[i] * y0[i] + m[i];
    });
}

int mpcx_aby3_vecbool_mul_local(mpcx_ctx *ctx, const int8_t *X0, const int8_t *X1, const int8_t *Y0, const int8_t *Y1,
                                uint64_t n, int8_t *Z0) {
    MPCX_REQUIRE((X0 && X1 && Y0 && Y1 && Z0) || n == 0, "mpcx_aby3_vecbool_mul_local: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { Z0[i] = (i8)((X0[i] & Y0[i]) ^ (X1[i] & Y0[i]) ^ (X0[i] & Y1[i])); });
}

int mpcx_aby3_reveal3(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, const int64_t *c, int64_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && c && out) || n == 0, "mpcx_aby3_reveal3: NULL pointer");
    const u64 *x = (const u64 *)a, *y = (const u64 *)b, *z = (const u64 *)c;
    u64 *o = (u64 *)out;
    return ew_launch(ctx, n, [=] __device__(u64 i) { o[i] = x[i] + y[i] + z[i]; });
}
int mpcx_aby3_reveal3_bools(mpcx_ctx *ctx, const int8_t *a, const int8_t *b, const int8_t *c, int8_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && c && out) || n == 0, "mpcx_aby3_reveal3_bools: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { out[i] = (i8)(a[i] ^ b[i] ^ c[i]); });
}
int mpcx_aby3_reveal3_bits(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, const int64_t *c, int64_t *out,
                           uint64_t n) {
    MPCX_REQUIRE((a && b && c && out) || n == 0, "mpcx_aby3_reveal3_bits: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { out[i] = a[i] ^ b[i] ^ c[i]; });
}

int mpcx_aby3_add_prepare_step1(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, uint64_t n, int64_t *C0) {
    MPCX_REQUIRE((X0 && X1 && C0) || n == 0, "mpcx_aby3_add_prepare_step1: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { C0[i] = X0[i] & X1[i]; });
}
int mpcx_aby3_add_prepare_step2(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *C0,
                                const int64_t *C1, uint64_t n, int64_t *P0, int64_t *P1, int64_t *G0) {
    MPCX_REQUIRE((X0 && X1 && C0 && C1 && P0 && P1 && G0) || n == 0, "mpcx_aby3_add_prepare_step2: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 s0 = (u64)X0[i], s1 = (u64)X1[i];
        const u64 c0 = (u64)C0[i] << 1, c1 = (u64)C1[i] << 1;
        P0[i] = (i64)(c0 ^ s0);
        P1[i] = (i64)(c1 ^ s1);
        G0[i] = (i64)((c0 & s0) ^ (c0 & s1) ^ (c1 & s0));
    });
}

int mpcx_aby3_msb_begin(mpcx_ctx *ctx, const int64_t *G0, const int64_t *G1, const int64_t *P0, const int64_t *P1,
                        uint64_t n, uint64_t *state) {
    if (n == 0) return 0;
    MPCX_REQUIRE(G0 && G1 && P0 && P1 && state, "mpcx_aby3_msb_begin: NULL pointer");
    // the first level is read in place: remember where it lives (host side, per state buffer)
    ctx->msb_inputs[state] = {G0, G1, P0, P1};
    return 0;
}
static int msb_round_check(mpcx_ctx *ctx, const char *who, const void *state, const void *msg, uint64_t temp) {
    MPCX_REQUIRE(state && msg, "%s: NULL pointer", who);
    MPCX_REQUIRE(temp == 32 || temp == 16 || temp == 8 || temp == 4 || temp == 2 || temp == 1,
                 "%s: temp must be one of 32,16,8,4,2,1", who);
    MPCX_REQUIRE(((uintptr_t)msg & 7) == 0, "%s: the message must be 8-byte aligned", who);
    return 0;
}
// the four arrays the round with `temp` pairs reads
static int msb_level_in(mpcx_ctx *ctx, const char *who, const uint64_t *state, uint64_t n, uint64_t temp, MsbIn *in) {
    if (temp == 32) {
        auto it = ctx->msb_inputs.find(state);
        MPCX_REQUIRE(it != ctx->msb_inputs.end(), "%s: mpcx_aby3_msb_begin was not called for this state buffer", who);
        for (int k = 0; k < 4; k++) in->a[k] = (const u32 *)it->second[k];
        return 0;
    }
    const u8 *base = (const u8 *)state + msb_base(n, (u32)temp);
    for (int k = 0; k < 4; k++) in->a[k] = (const u32 *)(base + k * msb_stride(n, (u32)temp));
    return 0;
}
// launches KERNEL<16-byte aligned message, first level> for the round with `temp` pairs
#define MSB_ROUND_LAUNCH_BITS(KERNEL, MSG)                                                                             \
    do {                                                                                                               \
        u8 *st = (u8 *)state;                                                                                          \
        const u32 T = (u32)temp;                                                                                       \
        const u64 nbits = 2 * temp * n;                                                                                \
        const dim3 grid(grid_for(ceil_div(nbits, 32))), block(256);                                                    \
        const u64 sout = msb_stride(n, T / 2);                                                                         \
        u8 *out = st + msb_base(n, T / 2);                                                                             \
        if (T == 32)                                                                                                   \
            hipLaunchKernelGGL((KERNEL<true, true, true>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA); \
        else                                                                                                           \
            hipLaunchKernelGGL((KERNEL<true, false, true>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA); \
    } while (0)
#define MSB_ROUND_LAUNCH(KERNEL, MSG)                                                                                  \
    do {                                                                                                               \
        u8 *st = (u8 *)state;                                                                                          \
        const u32 T = (u32)temp;                                                                                       \
        const u64 nbits = 2 * temp * n;                                                                                \
        const dim3 grid(grid_for(ceil_div(nbits, 32))), block(256);                                                    \
        const u64 sout = msb_stride(n, T / 2);                                                                         \
        u8 *out = st + msb_base(n, T / 2);                                                                             \
        const bool a16 = ((uintptr_t)(MSG) & 15) == 0;                                                                 \
        if (a16 && T == 32)                                                                                            \
            hipLaunchKernelGGL((KERNEL<true, true>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA);      \
        else if (a16)                                                                                                  \
            hipLaunchKernelGGL((KERNEL<true, false>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA);     \
        else if (T == 32)                                                                                              \
            hipLaunchKernelGGL((KERNEL<false, true>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA);     \
        else                                                                                                           \
            hipLaunchKernelGGL((KERNEL<false, false>), grid, block, 0, ctx->stream, in, out, sout, nbits, MSG EXTRA);    \
    } while (0)
int mpcx_aby3_msb_round_pre(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, int8_t *msg_out) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, "mpcx_aby3_msb_round_pre", state, msg_out, temp));
    MsbIn in;
    MPCX_TRY(msb_level_in(ctx, "mpcx_aby3_msb_round_pre", state, n, temp, &in));
    ProfScope ps(ctx, PROF_MSB);
    u8 *bk0 = (u8 *)state + msb_backup_base(n), *bk1 = bk0 + msb_backup_stride(n);
#define EXTRA , bk0, bk1
    MSB_ROUND_LAUNCH(msb_round_pre_kernel, (u8 *)msg_out);
#undef EXTRA
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_aby3_msb_round_pre_bits(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, uint32_t *bits_out) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, "mpcx_aby3_msb_round_pre_bits", state, bits_out, temp));
    MsbIn in;
    MPCX_TRY(msb_level_in(ctx, "mpcx_aby3_msb_round_pre_bits", state, n, temp, &in));
    ProfScope ps(ctx, PROF_MSB);
    u8 *bk0 = (u8 *)state + msb_backup_base(n), *bk1 = bk0 + msb_backup_stride(n);
#define EXTRA , bk0, bk1
    MSB_ROUND_LAUNCH_BITS(msb_round_pre_kernel, (u8 *)bits_out);
#undef EXTRA
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_aby3_msb_round_post(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, "mpcx_aby3_msb_round_post", state, msg_in, temp));
    MsbIn in;
    MPCX_TRY(msb_level_in(ctx, "mpcx_aby3_msb_round_post", state, n, temp, &in));
    ProfScope ps(ctx, PROF_MSB);
#define EXTRA
    MSB_ROUND_LAUNCH(msb_round_post_kernel, (const u8 *)msg_in);
#undef EXTRA
    MPCX_LAUNCH_CHECK();
    return 0;
}
#undef MSB_ROUND_LAUNCH
#undef MSB_ROUND_LAUNCH_BITS
// bits = 0: the messages are wire bytes; 1: bit strings (the _bits entry points)
static int msb_round_mid_impl(mpcx_ctx *ctx, const char *who, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in,
                              int8_t *msg_out, int bits) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, who, state, msg_in, temp));
    MPCX_REQUIRE(temp != 1, "%s: temp = 1 is the last round (mpcx_aby3_msb_last)", who);
    MPCX_REQUIRE(msg_out && ((uintptr_t)msg_out & 7) == 0, "%s: msg_out must be 8-byte aligned", who);
    const u32 T = (u32)temp;
    const u64 nbits = 2 * temp * n;
    if (nbits < 4096 && !bits) {
        // A handful of lanes: the word-rounded arrays of level T and level T/4 may share bytes of the state, which
        // the one-pass kernel would write while other lanes still read them.  Two passes, as the separate calls do.
        MPCX_TRY(mpcx_aby3_msb_round_post(ctx, state, n, temp, msg_in));
        return mpcx_aby3_msb_round_pre(ctx, state, n, temp / 2, msg_out);
    }
    if (nbits < 4096) { // the same two passes on a bit string: expand it on the device, then the packed first half
        void *ws;
        MPCX_TRY(mpcx_ws(ctx, (nbits + 15) & ~15ull, &ws));
        u8 *bytes = (u8 *)ws;
        const u32 *w = (const u32 *)msg_in;
        MPCX_TRY(ew_launch(ctx, nbits, [=] __device__(u64 t) { bytes[t] = (u8)((w[t >> 5] >> (t & 31)) & 1u); }));
        MPCX_TRY(mpcx_aby3_msb_round_post(ctx, state, n, temp, (const i8 *)bytes));
        return mpcx_aby3_msb_round_pre_bits(ctx, state, n, temp / 2, (u32 *)msg_out);
    }
    MsbIn in;
    MPCX_TRY(msb_level_in(ctx, "mpcx_aby3_msb_round_mid", state, n, temp, &in));
    u8 *st = (u8 *)state;
    u8 *half = st + msb_base(n, T / 2), *quarter = st + msb_base(n, T / 4);
    const u64 shalf = msb_stride(n, T / 2), squarter = msb_stride(n, T / 4);
    const u64 g1_words = T == 32 ? 2 * n : msb_stride(n, T) / 4;
    const dim3 grid(grid_for(ceil_div(nbits, 64))), block(256);
    const bool a16 = (((uintptr_t)msg_in | (uintptr_t)msg_out) & 15) == 0;
    const u8 *mi = (const u8 *)msg_in;
    u8 *mo = (u8 *)msg_out;
    ProfScope ps(ctx, PROF_MSB);
#define MSB_MID(A16, FIRST)                                                                                            \
    hipLaunchKernelGGL((msb_round_mid_kernel<A16, FIRST>), grid, block, 0, ctx->stream, in.a[1], g1_words, half, shalf,   \
                       quarter, squarter, nbits, mi, mo)
    if (bits) {
        if (T == 32)
            hipLaunchKernelGGL((msb_round_mid_kernel<true, true, true>), grid, block, 0, ctx->stream, in.a[1], g1_words, half,
                               shalf, quarter, squarter, nbits, mi, mo);
        else
            hipLaunchKernelGGL((msb_round_mid_kernel<true, false, true>), grid, block, 0, ctx->stream, in.a[1], g1_words, half,
                               shalf, quarter, squarter, nbits, mi, mo);
    } else if (a16 && T == 32)
        MSB_MID(true, true);
    else if (a16)
        MSB_MID(true, false);
    else if (T == 32)
        MSB_MID(false, true);
    else
        MSB_MID(false, false);
#undef MSB_MID
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_aby3_msb_round_mid(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in,
                            int8_t *msg_out) {
    return msb_round_mid_impl(ctx, "mpcx_aby3_msb_round_mid", state, n, temp, msg_in, msg_out, 0);
}
int mpcx_aby3_msb_round_mid_bits(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const uint32_t *bits_in,
                                 uint32_t *bits_out) {
    return msb_round_mid_impl(ctx, "mpcx_aby3_msb_round_mid_bits", state, n, temp, (const int8_t *)bits_in, (int8_t *)bits_out, 1);
}
int mpcx_aby3_msb_final(mpcx_ctx *ctx, const uint64_t *state, const int64_t *P0_in, const int64_t *P1_in, uint64_t n,
                        int8_t *Z0, int8_t *Z1) {
    MPCX_REQUIRE((state && P0_in && P1_in && Z0 && Z1) || n == 0, "mpcx_aby3_msb_final: NULL pointer");
    ctx->msb_inputs.erase(state);
    const u8 *g0 = (const u8 *)state + msb_base(n, 0), *g1 = g0 + msb_stride(n, 0);
    return ew_launch(ctx, n, [=] __device__(u64 i) { // level 0: one bit per element and array
        Z0[i] = (i8)(((g0[i >> 3] >> (i & 7)) & 1) ^ ((u64)P0_in[i] >> 63));
        Z1[i] = (i8)(((g1[i >> 3] >> (i & 7)) & 1) ^ ((u64)P1_in[i] >> 63));
    });
}
int mpcx_aby3_msb_last_bits(mpcx_ctx *ctx, const uint64_t *state, uint64_t n, const uint32_t *bits_in, int8_t *Z0, int8_t *Z1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(state && bits_in && Z0 && Z1, "mpcx_aby3_msb_last_bits: NULL pointer");
    ctx->msb_inputs.erase(state);
    const u8 *st = (const u8 *)state;
    const u8 *g1_lvl1 = st + msb_base(n, 1) + msb_stride(n, 1), *g0_lvl0 = st + msb_base(n, 0);
    ProfScope ps(ctx, PROF_MSB);
    const u8 *bk0 = st + msb_backup_base(n), *bk1 = bk0 + msb_backup_stride(n);
    const dim3 grid(grid_for(ceil_div(n, 8))), block(256);
    hipLaunchKernelGGL((msb_last_kernel<false, true>), grid, block, 0, ctx->stream, g1_lvl1, g0_lvl0, (const u8 *)bits_in, bk0,
                       bk1, n, (u8 *)Z0, (u8 *)Z1);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_aby3_msb_last(mpcx_ctx *ctx, const uint64_t *state, uint64_t n, const int8_t *msg_in, int8_t *Z0, int8_t *Z1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(state && msg_in && Z0 && Z1, "mpcx_aby3_msb_last: NULL pointer");
    ctx->msb_inputs.erase(state);
    const u8 *st = (const u8 *)state;
    const u8 *g1_lvl1 = st + msb_base(n, 1) + msb_stride(n, 1), *g0_lvl0 = st + msb_base(n, 0);
    ProfScope ps(ctx, PROF_MSB);
    const u8 *bk0 = st + msb_backup_base(n), *bk1 = bk0 + msb_backup_stride(n);
    const dim3 grid(grid_for(ceil_div(n, 8))), block(256);
    if ((((uintptr_t)msg_in & 15) | (((uintptr_t)Z0 | (uintptr_t)Z1) & 7)) == 0)
        hipLaunchKernelGGL(msb_last_kernel<true>, grid, block, 0, ctx->stream, g1_lvl1, g0_lvl0, (const u8 *)msg_in, bk0, bk1, n,
                           (u8 *)Z0, (u8 *)Z1);
    else
        hipLaunchKernelGGL(msb_last_kernel<false>, grid, block, 0, ctx->stream, g1_lvl1, g0_lvl0, (const u8 *)msg_in, bk0, bk1,
                           n, (u8 *)Z0, (u8 *)Z1);
    MPCX_LAUNCH_CHECK();
    return 0;
}

int mpcx_aby3_trunc_local(mpcx_ctx *ctx, int party, const int64_t *X0, const int64_t *X1, uint64_t n, int number,
                          uint64_t seed12, uint64_t pop12, int64_t *Y0) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party >= 0 && party <= 2, "mpcx_aby3_trunc_local: party %d", party);
    MPCX_REQUIRE(number >= 0 && number < 64, "mpcx_aby3_trunc_local: shift %d out of range", number);
    MPCX_REQUIRE(Y0 && (party == 2 || X0) && (party != 1 || X1), "mpcx_aby3_trunc_local: NULL pointer");
    if (party == 0) return ew_launch(ctx, n, [=] __device__(u64 i) { Y0[i] = X0[i] >> number; });
    if (party == 2) return prg_fill_words(ctx, seed12, 0, PRG_POP_T0 + pop12, n, 1, Y0, nullptr);
    return prg_map1(ctx, seed12, PRG_POP_T0 + pop12, n, [=] __device__(u64 i, u64 w) {
        i64 t = (i64)((u64)X0[i] + (u64)X1[i]);
        t >>= number;
        Y0[i] = (i64)((u64)t - w);
    });
}

// ---- bit_inject.  A call works on elements [i0, i0 + n) of a vector of n_total elements.  The reference lays the
// words of a stream out per whole vector (ABY3Engine.cpp:539-558): C[n_total], then n_total (W_0, W_1) pairs, and on
// s01 the same again for the second OT.  Every use of those words is fused with their regeneration (prg_map.h):
//   * masks C: one word per element; the two C words an element needs come from two streams in one pass;
//   * W pairs: element e owns words 2e, 2e+1 of its section; word index w is (e = w >> 1, t = w & 1).
// A sender's whole segment before the round is one pass (bi_sender_pre_kernel): nothing is staged in memory.
} // extern "C"
namespace {
struct BiGeom {
    u64 nt, i0, n;
    // stream word of C[i0] / of the pair of element i0, for the OT whose section starts `base` words into the stream
    u64 c(u64 base) const { return base + i0; }
    u64 w(u64 base) const { return base + nt + 2 * i0; }
};
using BiGen = PrgCursor; // a generator positioned anywhere in a stream (prg_map.h)
struct BiSenderArgs {
    const PrgCk *ckCa, *ckCb, *ckW1, *ckW2, *ckWs; // checkpoint tables of the five stream sections
    u64 tCa, tCb, tW1, tW2, tWs;                   // stream word of element 0 of the call in each section
    const u64 *A0, *A1;                            // a = A0 (+ A1 when non-null)
    const i8 *Bx, *By, *sel;                       // b_x ^ b_y picks the message, sel the helper word
    u64 *d1a, *d1b, *d2a, *d2b, *ds;               // outputs, n words each
    u64 n;
    int aligned; // every input array is 16-byte aligned
};
// The whole "before the round" segment of an OT sender of bit_inject (party 0: ABY3Engine.cpp:539-620, party 1:
// :709-790) in one pass, with nothing staged in memory.  For element e:
//     M_t = (t ^ b_x ^ b_y) a - (Ca[e] + Cb[e])                      t = 0, 1          (:547-548, :751-752)
//     d1a[e] = M_0 ^ W1_0[e], d1b[e] = M_1 ^ W1_1[e]     masked for the receiver, pair words of the helper's stream
//     d2a[e] = M_0 ^ W2_0[e], d2b[e] = M_1 ^ W2_1[e]     the same messages under the other OT's pair words
//     ds[e]  = Ws_{sel[e] == 1}[e]                       the word this party sends as the helper of the other OT
// A lane owns 64 consecutive elements and five generators (the two C sections advance one word per element, the
// three pair sections two), works through them in windows of 16 elements -- operands and bits are read from the
// lane's own contiguous 128 / 16 bytes -- and hands each output array of a window to the wavefront through LDS, which
// stores it in memory order (128-byte runs), as the other regenerating kernels do.
__global__ __launch_bounds__(256) void bi_sender_pre_kernel(BiSenderArgs P) {
    constexpr u32 EPL = 64, WIN = 16, PITCH = 2 * WIN + 1; // a row of the tile: one window of two output arrays
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 lane0 = (u64)blockIdx.x * 256; // first lane of the workgroup in the launch
    const u64 e0 = (lane0 + tid) * EPL;
    const bool active = e0 < P.n;
    BiGen ca, cb, w1, w2, ws;
    ca.open(P.ckCa, P.tCa + e0, active);
    cb.open(P.ckCb, P.tCb + e0, active);
    w1.open(P.ckW1, P.tW1 + 2 * e0, active);
    w2.open(P.ckW2, P.tW2 + 2 * e0, active);
    ws.open(P.ckWs, P.tWs + 2 * e0, active);
    u64 *row_tile = tile + tid * PITCH;
    // the wavefront stores half `h` of its lanes' rows of the tile at dst[element]
    auto flush = [&](u64 *__restrict__ dst, u32 win, u32 h) {
#pragma unroll 4
        for (u32 j = 0; j < WIN; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / WIN, col = idx % WIN;
            const u64 e = (lane0 + row) * EPL + win * WIN + col;
            if (e < P.n) dst[e] = tile[row * PITCH + h * WIN + col];
        }
    };
#pragma unroll 1
    for (u32 win = 0; win < EPL / WIN; win++) {
        const u64 eb = e0 + win * WIN;
        u64 m0[WIN], m1[WIN];
        u32 selbits = 0;
        if (P.aligned && eb + WIN <= P.n) { // the lane's 16 operands and bits are contiguous: 128 + 16 + 16 (+ 16) bytes
            const uint4 bx = *(const uint4 *)(P.Bx + eb), by = *(const uint4 *)(P.By + eb), sl = *(const uint4 *)(P.sel + eb);
            const u32 bxw[4] = {bx.x, bx.y, bx.z, bx.w}, byw[4] = {by.x, by.y, by.z, by.w}, slw[4] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                const ulonglong2 a01 = *(const ulonglong2 *)(P.A0 + eb + 4 * q), a23 = *(const ulonglong2 *)(P.A0 + eb + 4 * q + 2);
                u64 a[4] = {a01.x, a01.y, a23.x, a23.y};
                if (P.A1) {
                    const ulonglong2 c01 = *(const ulonglong2 *)(P.A1 + eb + 4 * q), c23 = *(const ulonglong2 *)(P.A1 + eb + 4 * q + 2);
                    a[0] += c01.x, a[1] += c01.y, a[2] += c23.x, a[3] += c23.y;
                }
#pragma unroll
                for (u32 e = 0; e < 4; e++) {
                    const u32 k = 4 * q + e;
                    const i64 t = (i64)(i8)(((bxw[q] ^ byw[q]) >> (8 * e)) & 0xFF);
                    selbits |= (u32)(((slw[q] >> (8 * e)) & 0xFF) == 1) << k;
                    m0[k] = (u64)(0 ^ t) * a[e];
                    m1[k] = (u64)(1 ^ t) * a[e];
                }
            }
        } else {
#pragma unroll
            for (u32 k = 0; k < WIN; k++) {
                u64 a = 0;
                i64 t = 0;
                if (eb + k < P.n) {
                    a = P.A0[eb + k];
                    if (P.A1) a += P.A1[eb + k];
                    t = (i64)(i8)(P.Bx[eb + k] ^ P.By[eb + k]);
                    selbits |= (u32)(P.sel[eb + k] == 1) << k;
                }
                m0[k] = (u64)(0 ^ t) * a;
                m1[k] = (u64)(1 ^ t) * a;
            }
        }
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            const u64 cc = ca.next() + cb.next();
            m0[k] -= cc;
            m1[k] -= cc;
        }
        // The sections' generators are independent of one another: the fences keep the scheduler from running them
        // side by side and holding all their words in registers.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            row_tile[k] = m0[k] ^ w1.next();
            row_tile[WIN + k] = m1[k] ^ w1.next();
        }
        prg_wave_sync();
        flush(P.d1a, win, 0);
        flush(P.d1b, win, 1);
        prg_wave_sync();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            row_tile[k] = m0[k] ^ w2.next();
            row_tile[WIN + k] = m1[k] ^ w2.next();
        }
        prg_wave_sync();
        flush(P.d2a, win, 0);
        flush(P.d2b, win, 1);
        prg_wave_sync();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            const u64 x0 = ws.next(), x1 = ws.next();
            row_tile[k] = ((selbits >> k) & 1) ? x1 : x0;
        }
        prg_wave_sync();
        flush(P.ds, win, 0);
        prg_wave_sync();
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The helper's segment before the round (party 2, ABY3Engine.cpp:875-922): da[e] = Wa_{sa[e] == 1}[e] and
// db[e] = Wb_{sb[e] == 1}[e] for the pair sections of two streams, in one pass with the geometry of
// bi_sender_pre_kernel (a lane owns 64 elements, windows of 16, stores in memory order through LDS).
struct BiHelperArgs {
    const PrgCk *ckWa, *ckWb;
    u64 tWa, tWb;
    const i8 *sa, *sb;
    u64 *da, *db;
    u64 n;
};
__global__ __launch_bounds__(256) void bi_helper_pre_kernel(BiHelperArgs P) {
    constexpr u32 EPL = 64, WIN = 16, PITCH = 2 * WIN + 1;
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 lane0 = (u64)blockIdx.x * 256;
    const u64 e0 = (lane0 + tid) * EPL;
    const bool active = e0 < P.n;
    BiGen wa, wb;
    wa.open(P.ckWa, P.tWa + 2 * e0, active);
    wb.open(P.ckWb, P.tWb + 2 * e0, active);
    u64 *row_tile = tile + tid * PITCH;
#pragma unroll 1
    for (u32 win = 0; win < EPL / WIN; win++) {
        const u64 eb = e0 + win * WIN;
        u32 abits = 0, bbits = 0;
#pragma unroll
        for (u32 k = 0; k < WIN; k++)
            if (eb + k < P.n) {
                abits |= (u32)(P.sa[eb + k] == 1) << k;
                bbits |= (u32)(P.sb[eb + k] == 1) << k;
            }
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            const u64 a0 = wa.next(), a1 = wa.next(), b0 = wb.next(), b1 = wb.next();
            row_tile[k] = ((abits >> k) & 1) ? a1 : a0;
            row_tile[WIN + k] = ((bbits >> k) & 1) ? b1 : b0;
        }
        prg_wave_sync();
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            u64 *__restrict__ dst = h ? P.db : P.da;
#pragma unroll 4
            for (u32 j = 0; j < WIN; j++) {
                const u32 idx = j * 64 + (tid & 63);
                const u32 row = (tid & ~63u) + idx / WIN, col = idx % WIN;
                const u64 e = (lane0 + row) * EPL + win * WIN + col;
                if (e < P.n) dst[e] = tile[row * PITCH + h * WIN + col];
            }
        }
        prg_wave_sync();
    }
}

// The segment after the round (party 0: ABY3Engine.cpp:621-682, party 1: :791-850, party 2: :923-1020) in one pass.
// Output share k of element e is
//     C_k[e] = G_k[e] + ((sel_k[e] == 1 ? hi_k[e] : lo_k[e]) ^ x_k[e])
// where G_k is one C word of a stream section or the sum of two (the shares the senders kept), and the second term
// -- the message the receiver unmasks with the helper's word (:660, :811, :963, :996) -- is present when sel_k is
// non-null.  With sub_k non-null the pass stores sub_k[e] - C_k[e] instead: the ReLU composition x - bit_inject(x, msb)
// (benchmark/benchmark1.cpp:71-73 followed by the subtraction) without writing C and reading it back.
// Geometry of bi_sender_pre_kernel: a lane owns 64 consecutive elements and the generators of both outputs, works in
// windows of 16 elements and hands one output of a window at a time to the wavefront through LDS; the wavefront
// visits it in memory order, so the message words, the select bytes and the stores run in 128-byte pieces.
struct BiPostArgs {
    const PrgCk *ck[2][2]; // generators of G_k: ck[k][0], plus ck[k][1] when non-null
    u64 t[2][2];           // stream word of element 0 of the call in each section
    const i8 *sel[2];
    const u64 *lo[2], *hi[2], *x[2];
    const u64 *sub[2];
    u64 *out[2];
    u64 n;
};
// one output of a window: TWO = G_k is the sum of two sections, SEL = the unmasked message is added, SUB = the minuend.
// Both candidate message words are loaded and one is chosen afterwards, so that no load waits for the select byte;
// the arrays are compile-time present or absent, which lets the loads of four elements run together.
// elements of a window (32 measured the same: the run length is not what bounds the pass)
#ifndef BI_POST_WIN
#define BI_POST_WIN 16
#endif
template <bool TWO, bool SEL, bool SUB, bool PAIRS>
__device__ inline void bi_post_pass(const BiPostArgs &P, const u32 k, BiGen &ga, BiGen &gb, u64 *tile, u64 *row_tile,
                                    const u32 tid, const u64 lane0, const u32 win) {
    constexpr u32 EPL = 64, WIN = BI_POST_WIN, PITCH = WIN + 1;
#pragma unroll
    for (u32 j = 0; j < WIN; j++) row_tile[j] = TWO ? ga.next() + gb.next() : ga.next();
    prg_wave_sync();
    const i8 *__restrict__ sel = P.sel[k];
    const u64 *__restrict__ lo = P.lo[k], *__restrict__ hi = P.hi[k], *__restrict__ x = P.x[k];
    const u64 *__restrict__ sub = P.sub[k];
    u64 *__restrict__ out = P.out[k];
    if (PAIRS) {
        // every array is 16-byte aligned: a lane finishes two neighbouring elements with 16-byte accesses
#pragma unroll 4
        for (u32 j = 0; j < WIN / 2; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / (WIN / 2), col = 2 * (idx % (WIN / 2));
            const u64 e = (lane0 + row) * EPL + win * WIN + col;
            if (e + 2 <= P.n) {
                u64 v0 = tile[row * PITCH + col], v1 = tile[row * PITCH + col + 1];
                if (SEL) {
                    const ulonglong2 m0 = *(const ulonglong2 *)(lo + e), m1 = *(const ulonglong2 *)(hi + e);
                    const ulonglong2 w = *(const ulonglong2 *)(x + e);
                    const u32 sb = *(const u16 *)(sel + e);
                    v0 += ((sb & 0xFF) == 1 ? m1.x : m0.x) ^ w.x;
                    v1 += ((sb >> 8) == 1 ? m1.y : m0.y) ^ w.y;
                }
                if (SUB) {
                    const ulonglong2 sv = *(const ulonglong2 *)(sub + e);
                    v0 = sv.x - v0, v1 = sv.y - v1;
                }
                *(ulonglong2 *)(out + e) = make_ulonglong2(v0, v1);
            } else if (e < P.n) { // the last element of an odd-length call
                u64 v = tile[row * PITCH + col];
                if (SEL) v += (sel[e] == 1 ? hi[e] : lo[e]) ^ x[e];
                if (SUB) v = sub[e] - v;
                out[e] = v;
            }
        }
    } else {
#pragma unroll 4
        for (u32 j = 0; j < WIN; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / WIN, col = idx % WIN;
            const u64 e = (lane0 + row) * EPL + win * WIN + col;
            if (e < P.n) {
                u64 v = tile[row * PITCH + col];
                if (SEL) {
                    const u64 m0 = lo[e], m1 = hi[e], w = x[e];
                    v += (sel[e] == 1 ? m1 : m0) ^ w;
                }
                if (SUB) v = sub[e] - v;
                out[e] = v;
            }
        }
    }
    prg_wave_sync();
}
// PARTY 0: output 0 = one section + message, output 1 = two sections; PARTY 1: the other way round; PARTY 2: both
// outputs one section + message.
template <int PARTY, bool SUB, bool PAIRS> __global__ __launch_bounds__(256) void bi_post_kernel(BiPostArgs P) {
    constexpr u32 EPL = 64, WIN = BI_POST_WIN, PITCH = WIN + 1;
    constexpr bool TWO0 = PARTY == 1, TWO1 = PARTY == 0;
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 lane0 = (u64)blockIdx.x * 256;
    const u64 e0 = (lane0 + tid) * EPL;
    const bool active = e0 < P.n;
    BiGen g[2][2];
    g[0][0].open(P.ck[0][0], P.t[0][0] + e0, active);
    if (TWO0) g[0][1].open(P.ck[0][1], P.t[0][1] + e0, active);
    g[1][0].open(P.ck[1][0], P.t[1][0] + e0, active);
    if (TWO1) g[1][1].open(P.ck[1][1], P.t[1][1] + e0, active);
    u64 *row_tile = tile + tid * PITCH;
#pragma unroll 1
    for (u32 win = 0; win < EPL / WIN; win++) {
        bi_post_pass<TWO0, !TWO0, SUB, PAIRS>(P, 0, g[0][0], g[0][1], tile, row_tile, tid, lane0, win);
        bi_post_pass<TWO1, !TWO1, SUB, PAIRS>(P, 1, g[1][0], g[1][1], tile, row_tile, tid, lane0, win);
    }
}
int bi_post_launch(mpcx_ctx *ctx, int party, const BiPostArgs &P) {
    const u64 grid = ceil_div(ceil_div(P.n, 64), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "bit_inject: range too large for one launch");
    const bool sub = P.sub[0] != nullptr;
    uintptr_t bits = 0; // two elements per lane and access when every array in use starts on a 16-byte boundary
    for (int k = 0; k < 2; k++)
        bits |= (uintptr_t)P.lo[k] | (uintptr_t)P.hi[k] | (uintptr_t)P.x[k] | (uintptr_t)P.sub[k] | (uintptr_t)P.out[k] |
                ((uintptr_t)P.sel[k] & 1) * 8;
    const bool pairs = (bits & 15) == 0;
    ProfScope ps(ctx, PROF_PRG);
#define BI_POST_LAUNCH(PARTY, SUBF, PAIRSF)                                                                             \
    hipLaunchKernelGGL((bi_post_kernel<PARTY, SUBF, PAIRSF>), dim3((u32)grid), dim3(256), 0, ctx->stream, P)
#define BI_POST(PARTY)                                                                                                 \
    do {                                                                                                               \
        if (sub && pairs)                                                                                              \
            BI_POST_LAUNCH(PARTY, true, true);                                                                         \
        else if (sub)                                                                                                  \
            BI_POST_LAUNCH(PARTY, true, false);                                                                        \
        else if (pairs)                                                                                                \
            BI_POST_LAUNCH(PARTY, false, true);                                                                        \
        else                                                                                                           \
            BI_POST_LAUNCH(PARTY, false, false);                                                                       \
    } while (0)
    if (party == 0) {
        BI_POST(0);
    } else if (party == 1) {
        BI_POST(1);
    } else {
        BI_POST(2);
    }
#undef BI_POST_LAUNCH
#undef BI_POST
    MPCX_LAUNCH_CHECK();
    return 0;
}

int bi_sender_launch(mpcx_ctx *ctx, BiSenderArgs P) {
    P.aligned = (((uintptr_t)P.A0 | (uintptr_t)P.A1 | (uintptr_t)P.Bx | (uintptr_t)P.By | (uintptr_t)P.sel) & 15) == 0;
    const u64 grid = ceil_div(ceil_div(P.n, 64), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "bit_inject: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(bi_sender_pre_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, P);
    MPCX_LAUNCH_CHECK();
    return 0;
}
} // namespace
#define BI_RANGE_CHECK(name)                                                                                           \
    MPCX_REQUIRE(i0 <= n_total && n <= n_total - i0, name ": elements [%llu, +%llu) outside a vector of %llu",          \
                 (unsigned long long)i0, (unsigned long long)n, (unsigned long long)n_total)
extern "C" {
int mpcx_aby3_bit_inject_p0_pre(mpcx_ctx *ctx, const int64_t *A0, const int8_t *B0, const int8_t *B1, uint64_t n_total,
                                uint64_t i0, uint64_t n, uint64_t seed01, uint64_t pop01, uint64_t seed20,
                                uint64_t pop20, int64_t *to_p1, int64_t *to_p2) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p0_pre");
    if (n == 0) return 0;
    MPCX_REQUIRE(A0 && B0 && B1 && to_p1 && to_p2, "mpcx_aby3_bit_inject_p0_pre: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b20 = PRG_POP_T0 + pop20;
    const PrgCk *ck01, *ck20;
    MPCX_TRY(mpcx_prg_table(ctx, seed01, 0, b01 + 6 * n_total, &ck01));
    MPCX_TRY(mpcx_prg_table(ctx, seed20, 0, b20 + 3 * n_total, &ck20));
    BiSenderArgs P;
    P.ckCa = ck01, P.tCa = G.c(b01);                               // C01
    P.ckCb = ck20, P.tCb = G.c(b20);                               // C02
    P.ckW1 = ck20, P.tW1 = G.w(b20);                               // W02: the messages for party 1 (:590-599)
    P.ckW2 = ck01, P.tW2 = G.w(b01);                               // W01: the same messages for party 2
    P.ckWs = ck01, P.tWs = G.w(b01 + 3 * n_total);                 // W10_b (:603)
    P.A0 = (const u64 *)A0, P.A1 = nullptr, P.Bx = B0, P.By = B1, P.sel = B0;
    P.d1a = (u64 *)to_p1, P.d1b = (u64 *)to_p1 + n;
    P.d2a = (u64 *)to_p2, P.d2b = (u64 *)to_p2 + n, P.ds = (u64 *)to_p2 + 2 * n;
    P.n = n;
    return bi_sender_launch(ctx, P);
}
static int bi_p0_post(mpcx_ctx *ctx, const char *who, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                      uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20, const int64_t *from_p1,
                      const int64_t *from_p2, const int64_t *S0, const int64_t *S1, int64_t *C0, int64_t *C1) {
    MPCX_REQUIRE(i0 <= n_total && n <= n_total - i0, "%s: elements [%llu, +%llu) outside a vector of %llu", who,
                 (unsigned long long)i0, (unsigned long long)n, (unsigned long long)n_total);
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && from_p1 && from_p2 && C0 && C1, "%s: NULL pointer", who);
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b20 = PRG_POP_T0 + pop20;
    const PrgCk *ck01, *ck20;
    MPCX_TRY(mpcx_prg_table(ctx, seed01, 0, b01 + 6 * n_total, &ck01));
    MPCX_TRY(mpcx_prg_table(ctx, seed20, 0, b20 + 3 * n_total, &ck20));
    const u64 *f1 = (const u64 *)from_p1, *f2 = (const u64 *)from_p2;
    BiPostArgs P{};
    // A0j[0] + A12j[0] = C02 + (m_b ^ w_b)  (:660, :679)
    P.ck[0][0] = ck20, P.t[0][0] = G.c(b20);
    P.sel[0] = B0, P.lo[0] = f1, P.hi[0] = f1 + n, P.x[0] = f2;
    // A0j[1] + A12j[1] = C01 + C10
    P.ck[1][0] = ck01, P.t[1][0] = G.c(b01);
    P.ck[1][1] = ck01, P.t[1][1] = G.c(b01 + 3 * n_total);
    P.sub[0] = (const u64 *)S0, P.sub[1] = (const u64 *)S1;
    P.out[0] = (u64 *)C0, P.out[1] = (u64 *)C1, P.n = n;
    return bi_post_launch(ctx, 0, P);
}
int mpcx_aby3_bit_inject_p0_post(mpcx_ctx *ctx, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20,
                                 const int64_t *from_p1, const int64_t *from_p2, int64_t *C0, int64_t *C1) {
    return bi_p0_post(ctx, "mpcx_aby3_bit_inject_p0_post", B0, n_total, i0, n, seed01, pop01, seed20, pop20, from_p1, from_p2,
                      nullptr, nullptr, C0, C1);
}
int mpcx_aby3_bit_inject_p0_post_sub(mpcx_ctx *ctx, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                                     uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20,
                                     const int64_t *from_p1, const int64_t *from_p2, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1) {
    MPCX_REQUIRE((S0 && S1) || n == 0, "mpcx_aby3_bit_inject_p0_post_sub: NULL minuend");
    return bi_p0_post(ctx, "mpcx_aby3_bit_inject_p0_post_sub", B0, n_total, i0, n, seed01, pop01, seed20, pop20, from_p1,
                      from_p2, S0, S1, R0, R1);
}
int mpcx_aby3_bit_inject_p1_pre(mpcx_ctx *ctx, const int64_t *A0, const int64_t *A1, const int8_t *B0,
                                const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                                uint64_t pop01, uint64_t seed12, uint64_t pop12, int64_t *to_p0, int64_t *to_p2) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p1_pre");
    if (n == 0) return 0;
    MPCX_REQUIRE(A0 && A1 && B0 && B1 && to_p0 && to_p2, "mpcx_aby3_bit_inject_p1_pre: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b12 = PRG_POP_T0 + pop12;
    const PrgCk *ck01, *ck12;
    MPCX_TRY(mpcx_prg_table(ctx, seed01, 0, b01 + 6 * n_total, &ck01));
    MPCX_TRY(mpcx_prg_table(ctx, seed12, 0, b12 + 3 * n_total, &ck12));
    BiSenderArgs P;
    P.ckCa = ck01, P.tCa = G.c(b01 + 3 * n_total);                 // C10
    P.ckCb = ck12, P.tCb = G.c(b12);                               // C12
    P.ckW1 = ck12, P.tW1 = G.w(b12);                               // W12: the messages for party 0 (:743-761)
    P.ckW2 = ck01, P.tW2 = G.w(b01 + 3 * n_total);                 // W10: the same messages for party 2
    P.ckWs = ck01, P.tWs = G.w(b01);                               // W01_b (:722)
    P.A0 = (const u64 *)A0, P.A1 = (const u64 *)A1, P.Bx = B0, P.By = B1, P.sel = B1;
    P.d1a = (u64 *)to_p0, P.d1b = (u64 *)to_p0 + n;
    P.d2a = (u64 *)to_p2 + n, P.d2b = (u64 *)to_p2 + 2 * n, P.ds = (u64 *)to_p2;
    P.n = n;
    return bi_sender_launch(ctx, P);
}
static int bi_p1_post(mpcx_ctx *ctx, const char *who, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                      uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12, const int64_t *from_p0,
                      const int64_t *from_p2, const int64_t *S0, const int64_t *S1, int64_t *C0, int64_t *C1) {
    MPCX_REQUIRE(i0 <= n_total && n <= n_total - i0, "%s: elements [%llu, +%llu) outside a vector of %llu", who,
                 (unsigned long long)i0, (unsigned long long)n, (unsigned long long)n_total);
    if (n == 0) return 0;
    MPCX_REQUIRE(B1 && from_p0 && from_p2 && C0 && C1, "%s: NULL pointer", who);
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b12 = PRG_POP_T0 + pop12;
    const PrgCk *ck01, *ck12;
    MPCX_TRY(mpcx_prg_table(ctx, seed01, 0, b01 + 6 * n_total, &ck01));
    MPCX_TRY(mpcx_prg_table(ctx, seed12, 0, b12 + 3 * n_total, &ck12));
    const u64 *f0 = (const u64 *)from_p0, *f2 = (const u64 *)from_p2;
    BiPostArgs P{};
    // A0j[0] + A12j[0] = C01 + C10
    P.ck[0][0] = ck01, P.t[0][0] = G.c(b01);
    P.ck[0][1] = ck01, P.t[0][1] = G.c(b01 + 3 * n_total);
    // A0j[1] + A12j[1] = (m_b ^ w_b) + C12  (:811)
    P.ck[1][0] = ck12, P.t[1][0] = G.c(b12);
    P.sel[1] = B1, P.lo[1] = f0, P.hi[1] = f0 + n, P.x[1] = f2;
    P.sub[0] = (const u64 *)S0, P.sub[1] = (const u64 *)S1;
    P.out[0] = (u64 *)C0, P.out[1] = (u64 *)C1, P.n = n;
    return bi_post_launch(ctx, 1, P);
}
int mpcx_aby3_bit_inject_p1_post(mpcx_ctx *ctx, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p2, int64_t *C0, int64_t *C1) {
    return bi_p1_post(ctx, "mpcx_aby3_bit_inject_p1_post", B1, n_total, i0, n, seed01, pop01, seed12, pop12, from_p0, from_p2,
                      nullptr, nullptr, C0, C1);
}
int mpcx_aby3_bit_inject_p1_post_sub(mpcx_ctx *ctx, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                                     uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12,
                                     const int64_t *from_p0, const int64_t *from_p2, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1) {
    MPCX_REQUIRE((S0 && S1) || n == 0, "mpcx_aby3_bit_inject_p1_post_sub: NULL minuend");
    return bi_p1_post(ctx, "mpcx_aby3_bit_inject_p1_post_sub", B1, n_total, i0, n, seed01, pop01, seed12, pop12, from_p0,
                      from_p2, S0, S1, R0, R1);
}
int mpcx_aby3_bit_inject_p2_pre(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                int64_t *to_p0, int64_t *to_p1) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p2_pre");
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && B1 && to_p0 && to_p1, "mpcx_aby3_bit_inject_p2_pre: NULL pointer");
    const BiGeom G{n_total, i0, n};
    BiHelperArgs P;
    MPCX_TRY(mpcx_prg_table(ctx, seed20, 0, PRG_POP_T0 + pop20 + 3 * n_total, &P.ckWa));
    MPCX_TRY(mpcx_prg_table(ctx, seed12, 0, PRG_POP_T0 + pop12 + 3 * n_total, &P.ckWb));
    P.tWa = G.w(PRG_POP_T0 + pop20), P.sa = B0, P.da = (u64 *)to_p1; // W02_b (:888)
    P.tWb = G.w(PRG_POP_T0 + pop12), P.sb = B1, P.db = (u64 *)to_p0; // W12_b (:922)
    P.n = n;
    const u64 grid = ceil_div(ceil_div(n, 64), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "bit_inject: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(bi_helper_pre_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, P);
    MPCX_LAUNCH_CHECK();
    return 0;
}
static int bi_p2_post(mpcx_ctx *ctx, const char *who, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                      uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                      const int64_t *from_p0, const int64_t *from_p1, const int64_t *S0, const int64_t *S1, int64_t *C0,
                      int64_t *C1) {
    MPCX_REQUIRE(i0 <= n_total && n <= n_total - i0, "%s: elements [%llu, +%llu) outside a vector of %llu", who,
                 (unsigned long long)i0, (unsigned long long)n, (unsigned long long)n_total);
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && B1 && from_p0 && from_p1 && C0 && C1, "%s: NULL pointer", who);
    const BiGeom G{n_total, i0, n};
    const u64 b20 = PRG_POP_T0 + pop20, b12 = PRG_POP_T0 + pop12;
    const PrgCk *ck20, *ck12;
    MPCX_TRY(mpcx_prg_table(ctx, seed20, 0, b20 + 3 * n_total, &ck20));
    MPCX_TRY(mpcx_prg_table(ctx, seed12, 0, b12 + 3 * n_total, &ck12));
    const u64 *f0 = (const u64 *)from_p0, *f1 = (const u64 *)from_p1;
    BiPostArgs P{};
    // A0j[0] + A12j[0] = (m_b ^ w_b) + C12 with party 0's messages and party 1's word  (:963)
    P.ck[0][0] = ck12, P.t[0][0] = G.c(b12);
    P.sel[0] = B0, P.lo[0] = f0, P.hi[0] = f0 + n, P.x[0] = f1;
    // A0j[1] + A12j[1] = C02 + (m_b ^ w_b) with party 1's messages and party 0's word  (:996)
    P.ck[1][0] = ck20, P.t[1][0] = G.c(b20);
    P.sel[1] = B1, P.lo[1] = f1 + n, P.hi[1] = f1 + 2 * n, P.x[1] = f0 + 2 * n;
    P.sub[0] = (const u64 *)S0, P.sub[1] = (const u64 *)S1;
    P.out[0] = (u64 *)C0, P.out[1] = (u64 *)C1, P.n = n;
    return bi_post_launch(ctx, 2, P);
}
int mpcx_aby3_bit_inject_p2_post(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                 uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p1, int64_t *C0, int64_t *C1) {
    return bi_p2_post(ctx, "mpcx_aby3_bit_inject_p2_post", B0, B1, n_total, i0, n, seed20, pop20, seed12, pop12, from_p0,
                      from_p1, nullptr, nullptr, C0, C1);
}
int mpcx_aby3_bit_inject_p2_post_sub(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                     uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                     const int64_t *from_p0, const int64_t *from_p1, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1) {
    MPCX_REQUIRE((S0 && S1) || n == 0, "mpcx_aby3_bit_inject_p2_post_sub: NULL minuend");
    return bi_p2_post(ctx, "mpcx_aby3_bit_inject_p2_post_sub", B0, B1, n_total, i0, n, seed20, pop20, seed12, pop12, from_p0,
                      from_p1, S0, S1, R0, R1);
}

} // extern "C"
