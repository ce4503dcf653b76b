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
// word i of share array `arr` of the level a round reads.  FIRST (level 32): `in` holds the addresses of the
// caller's G0, G1, P0, P1; bit 63 of every element (bit 31 of the odd words) is forced to 0 in G and 1 in P.
template <bool FIRST> __device__ inline u32 msb_in_word(const u8 *in, u64 sin, u32 arr, u64 i) {
    if (!FIRST) return ((const u32 *)(in + arr * sin))[i];
    const u32 v = ((const u32 *)((const u64 *)in)[arr])[i];
    if (!(i & 1)) return v;
    return arr < 2 ? v & 0x7FFFFFFFu : v | 0x80000000u;
}

template <bool ALIGNED16, bool FIRST>
__global__ __launch_bounds__(256) void msb_round_pre_kernel(const u8 *__restrict__ in, u64 sin, u8 *__restrict__ out,
                                                            u64 sout, u64 nbits, u8 *__restrict__ msg) {
    constexpr u32 E = 0x55555555u;
    const u64 words = (nbits + 31) / 32;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) {
        const u32 g0 = msb_in_word<FIRST>(in, sin, 0, i), g1 = msb_in_word<FIRST>(in, sin, 1, i);
        const u32 p0 = msb_in_word<FIRST>(in, sin, 2, i), p1 = msb_in_word<FIRST>(in, sin, 3, i);
        const u32 po0 = (p0 >> 1) & E, po1 = (p1 >> 1) & E;
        const u32 ge0 = g0 & E, ge1 = g1 & E, pe0 = p0 & E, pe1 = p1 & E;
        const u32 re = (po0 & ge0) ^ (po1 & ge0) ^ (po0 & ge1);
        const u32 ro = (po0 & pe0) ^ (po1 & pe0) ^ (po0 & pe1);
        const u32 r = re | (ro << 1);
        u32 w[8];
#pragma unroll
        for (u32 q = 0; q < 8; q++) w[q] = nibble_to_bytes(r >> (4 * q));
        u8 *dst = msg + 32 * i;
        if (32 * i + 32 <= nbits) {
            if (ALIGNED16) {
                *(uint4 *)dst = make_uint4(w[0], w[1], w[2], w[3]);
                *(uint4 *)(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
            } else {
#pragma unroll
                for (u32 q = 0; q < 4; q++) *(uint2 *)(dst + 8 * q) = make_uint2(w[2 * q], w[2 * q + 1]);
            }
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
__global__ __launch_bounds__(256) void msb_round_post_kernel(const u8 *__restrict__ in, u64 sin, u8 *__restrict__ out,
                                                             u64 sout, u64 nbits, const u8 *__restrict__ msg) {
    constexpr u32 E = 0x55555555u;
    const u64 words = (nbits + 31) / 32;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < words; i += (u64)gridDim.x * 256) {
        const u8 *src = msg + 32 * i;
        u32 r = 0;
        if (32 * i + 32 <= nbits) {
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
#pragma unroll
            for (u32 q = 0; q < 8; q++) r |= bytes_to_nibble(w[q]) << (4 * q);
        } else {
            for (u64 t = 32 * i; t < nbits; t++) r |= (u32)(msg[t] & 1) << (t & 31);
        }
        const u32 g1 = msb_in_word<FIRST>(in, sin, 1, i);
        if (2 * i + 2 <= sout) {
            *(u16 *)(out + sout + 2 * i) = (u16)compress_even((r & E) ^ ((g1 >> 1) & E));
            *(u16 *)(out + 3 * sout + 2 * i) = (u16)compress_even((r >> 1) & E);
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
                            u64 c, int layout, u64 first_word, u64 seed_next, u64 pop_next, u64 seed_prev, u64 pop_prev,
                            i64 *Z0) {
    // the mask words go straight into Z0 and the product is accumulated onto them
    MPCX_TRY(prg_fill_diff(ctx, seed_next, PRG_POP_T0 + pop_next + first_word, seed_prev, PRG_POP_T0 + pop_prev + first_word,
                           rows * c, nullptr, Z0));
    GemmPart parts[2] = {{X0, Y0, Y1}, {X1, Y0, nullptr}};
    return ring_gemm_parts(ctx, parts, 2, rows, l, c, layout, layout, nullptr, nullptr, Z0, 0, 1);
}

int mpcx_aby3_matmul_local_rows(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0,
                                const int64_t *Y1, uint64_t rows, uint64_t l, uint64_t c, uint64_t row0,
                                uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                                int64_t *Z0) {
    MPCX_REQUIRE(X0 && X1 && Y0 && Y1 && Z0, "mpcx_aby3_matmul_local_rows: NULL operand");
    MPCX_REQUIRE(rows && l && c, "mpcx_aby3_matmul_local_rows: empty shape");
    return aby3_matmul_impl(ctx, X0, X1, Y0, Y1, rows, l, c, 0, row0 * c, seed_next, pop_next, seed_prev, pop_prev, Z0);
}

int mpcx_aby3_matmul_local(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                           uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next, uint64_t pop_next,
                           uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0) {
    MPCX_REQUIRE(X0 && X1 && Y0 && Y1 && Z0, "mpcx_aby3_matmul_local: NULL operand");
    MPCX_REQUIRE(r && l && c, "mpcx_aby3_matmul_local: empty shape");
    MPCX_REQUIRE(layout == 0 || layout == 1, "mpcx_aby3_matmul_local: layout must be 0 or 1");
    return aby3_matmul_impl(ctx, X0, X1, Y0, Y1, r, l, c, layout, 0, seed_next, pop_next, seed_prev, pop_prev, Z0);
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
    const u64 a0 = (u64)(uintptr_t)G0, a1 = (u64)(uintptr_t)G1, a2 = (u64)(uintptr_t)P0, a3 = (u64)(uintptr_t)P1;
    return ew_launch(ctx, 1, [=] __device__(u64) { state[0] = a0, state[1] = a1, state[2] = a2, state[3] = a3; });
}
static int msb_round_check(mpcx_ctx *ctx, const char *who, const void *state, const void *msg, uint64_t temp) {
    MPCX_REQUIRE(state && msg, "%s: NULL pointer", who);
    MPCX_REQUIRE(temp == 32 || temp == 16 || temp == 8 || temp == 4 || temp == 2 || temp == 1,
                 "%s: temp must be one of 32,16,8,4,2,1", who);
    MPCX_REQUIRE(((uintptr_t)msg & 7) == 0, "%s: the message must be 8-byte aligned", who);
    return 0;
}
// launches KERNEL<16-byte aligned message, first level> for the round with `temp` pairs
#define MSB_ROUND_LAUNCH(KERNEL, MSG)                                                                                  \
    do {                                                                                                               \
        u8 *st = (u8 *)state;                                                                                          \
        const u32 T = (u32)temp;                                                                                       \
        const u64 nbits = 2 * temp * n;                                                                                \
        const dim3 grid(grid_for(ceil_div(nbits, 32))), block(256);                                                    \
        const u8 *in = T == 32 ? st : st + msb_base(n, T);                                                             \
        const u64 sin = msb_stride(n, T), sout = msb_stride(n, T / 2);                                                 \
        u8 *out = st + msb_base(n, T / 2);                                                                             \
        const bool a16 = ((uintptr_t)(MSG) & 15) == 0;                                                                 \
        if (a16 && T == 32)                                                                                            \
            hipLaunchKernelGGL((KERNEL<true, true>), grid, block, 0, ctx->stream, in, sin, out, sout, nbits, MSG);       \
        else if (a16)                                                                                                  \
            hipLaunchKernelGGL((KERNEL<true, false>), grid, block, 0, ctx->stream, in, sin, out, sout, nbits, MSG);      \
        else if (T == 32)                                                                                              \
            hipLaunchKernelGGL((KERNEL<false, true>), grid, block, 0, ctx->stream, in, sin, out, sout, nbits, MSG);      \
        else                                                                                                           \
            hipLaunchKernelGGL((KERNEL<false, false>), grid, block, 0, ctx->stream, in, sin, out, sout, nbits, MSG);     \
    } while (0)
int mpcx_aby3_msb_round_pre(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, int8_t *msg_out) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, "mpcx_aby3_msb_round_pre", state, msg_out, temp));
    ProfScope ps(ctx, PROF_MSB);
    MSB_ROUND_LAUNCH(msb_round_pre_kernel, (u8 *)msg_out);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_aby3_msb_round_post(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in) {
    if (n == 0) return 0;
    MPCX_TRY(msb_round_check(ctx, "mpcx_aby3_msb_round_post", state, msg_in, temp));
    ProfScope ps(ctx, PROF_MSB);
    MSB_ROUND_LAUNCH(msb_round_post_kernel, (const u8 *)msg_in);
    MPCX_LAUNCH_CHECK();
    return 0;
}
#undef MSB_ROUND_LAUNCH
int mpcx_aby3_msb_final(mpcx_ctx *ctx, const uint64_t *state, const int64_t *P0_in, const int64_t *P1_in, uint64_t n,
                        int8_t *Z0, int8_t *Z1) {
    MPCX_REQUIRE((state && P0_in && P1_in && Z0 && Z1) || n == 0, "mpcx_aby3_msb_final: NULL pointer");
    const u8 *g0 = (const u8 *)state + msb_base(n, 0), *g1 = g0 + msb_stride(n, 0);
    return ew_launch(ctx, n, [=] __device__(u64 i) { // level 0: one bit per element and array
        Z0[i] = (i8)(((g0[i >> 3] >> (i & 7)) & 1) ^ ((u64)P0_in[i] >> 63));
        Z1[i] = (i8)(((g1[i >> 3] >> (i & 7)) & 1) ^ ((u64)P1_in[i] >> 63));
    });
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
// The two plain OT messages M_0, M_1 of a sender go through 16*n bytes of scratch between the passes.
} // extern "C"
namespace {
struct BiGeom {
    u64 nt, i0, n;
    // stream word of C[i0] / of the pair of element i0, for the OT whose section starts `base` words into the stream
    u64 c(u64 base) const { return base + i0; }
    u64 w(u64 base) const { return base + nt + 2 * i0; }
};
// out[t * n + e] = M[t * n + e] ^ W_t[e]: both masked OT messages of a sender from one pass over its pair words
int bi_mask_pairs(mpcx_ctx *ctx, u64 seed, u64 t_first, u64 n, const u64 *M, u64 *out) {
    return prg_map1(ctx, seed, t_first, 2 * n, [=] __device__(u64 w, u64 v) {
        const u64 k = (w & 1) * n + (w >> 1);
        out[k] = M[k] ^ v;
    });
}
// out[e] = W_b[e] with b = (sel[e] == 1): the helper's word for the receiver
int bi_select_pairs(mpcx_ctx *ctx, u64 seed, u64 t_first, u64 n, const i8 *sel, u64 *out) {
    return prg_map1(ctx, seed, t_first, 2 * n, [=] __device__(u64 w, u64 v) {
        const u64 e = w >> 1;
        if ((u64)(sel[e] == 1) == (w & 1)) out[e] = v;
    });
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
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, 16 * n, &ws));
    u64 *M = (u64 *)ws;
    // M_t = (t ^ b_x ^ b_y) a - c - c' with c + c' = C01 + C02 (ABY3Engine.cpp:547-548)
    MPCX_TRY(prg_map2(ctx, seed01, G.c(b01), seed20, G.c(b20), n, [] __device__(u64 a, u64 b) { return a + b; },
                      [=] __device__(u64 i, u64 cc) {
                          const u64 a = (u64)A0[i];
                          const i64 t = (i64)(B0[i] ^ B1[i]);
                          M[i] = (u64)(0 ^ t) * a - cc;
                          M[n + i] = (u64)(1 ^ t) * a - cc;
                      }));
    MPCX_TRY(bi_mask_pairs(ctx, seed20, G.w(b20), n, M, (u64 *)to_p1));                  // ^ W02
    MPCX_TRY(bi_mask_pairs(ctx, seed01, G.w(b01), n, M, (u64 *)to_p2));                  // ^ W01
    return bi_select_pairs(ctx, seed01, G.w(b01 + 3 * n_total), n, B0, (u64 *)to_p2 + 2 * n); // W10_b (:603)
}
int mpcx_aby3_bit_inject_p0_post(mpcx_ctx *ctx, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20,
                                 const int64_t *from_p1, const int64_t *from_p2, int64_t *C0, int64_t *C1) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p0_post");
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && from_p1 && from_p2 && C0 && C1, "mpcx_aby3_bit_inject_p0_post: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b20 = PRG_POP_T0 + pop20;
    const u64 *f1 = (const u64 *)from_p1, *f2 = (const u64 *)from_p2;
    u64 *o0 = (u64 *)C0, *o1 = (u64 *)C1;
    MPCX_TRY(prg_map1(ctx, seed20, G.c(b20), n, [=] __device__(u64 i, u64 c02) {
        const u64 a12 = (B0[i] == 1 ? f1[n + i] : f1[i]) ^ f2[i]; // :660
        o0[i] = c02 + a12;                                        // A0j[0] + A12j[0]
    }));
    return prg_map2(ctx, seed01, G.c(b01), seed01, G.c(b01 + 3 * n_total), n, [] __device__(u64 a, u64 b) { return a + b; },
                    [=] __device__(u64 i, u64 v) { o1[i] = v; }); // A0j[1] + A12j[1] = C01 + C10
}
int mpcx_aby3_bit_inject_p1_pre(mpcx_ctx *ctx, const int64_t *A0, const int64_t *A1, const int8_t *B0,
                                const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                                uint64_t pop01, uint64_t seed12, uint64_t pop12, int64_t *to_p0, int64_t *to_p2) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p1_pre");
    if (n == 0) return 0;
    MPCX_REQUIRE(A0 && A1 && B0 && B1 && to_p0 && to_p2, "mpcx_aby3_bit_inject_p1_pre: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b12 = PRG_POP_T0 + pop12;
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, 16 * n, &ws));
    u64 *M = (u64 *)ws;
    // M_t = (t ^ b_x ^ b_y)(a_1 + a_2) - C10 - C12 (:751-752)
    MPCX_TRY(prg_map2(ctx, seed01, G.c(b01 + 3 * n_total), seed12, G.c(b12), n, [] __device__(u64 a, u64 b) { return a + b; },
                      [=] __device__(u64 i, u64 cc) {
                          const u64 a = (u64)A0[i] + (u64)A1[i];
                          const i64 t = (i64)(B0[i] ^ B1[i]);
                          M[i] = (u64)(0 ^ t) * a - cc;
                          M[n + i] = (u64)(1 ^ t) * a - cc;
                      }));
    MPCX_TRY(bi_select_pairs(ctx, seed01, G.w(b01), n, B1, (u64 *)to_p2));                        // W01_b (:722)
    MPCX_TRY(bi_mask_pairs(ctx, seed01, G.w(b01 + 3 * n_total), n, M, (u64 *)to_p2 + n));         // ^ W10
    return bi_mask_pairs(ctx, seed12, G.w(b12), n, M, (u64 *)to_p0);                              // ^ W12
}
int mpcx_aby3_bit_inject_p1_post(mpcx_ctx *ctx, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p2, int64_t *C0, int64_t *C1) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p1_post");
    if (n == 0) return 0;
    MPCX_REQUIRE(B1 && from_p0 && from_p2 && C0 && C1, "mpcx_aby3_bit_inject_p1_post: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 b01 = PRG_POP_T0 + pop01, b12 = PRG_POP_T0 + pop12;
    const u64 *f0 = (const u64 *)from_p0, *f2 = (const u64 *)from_p2;
    u64 *o0 = (u64 *)C0, *o1 = (u64 *)C1;
    MPCX_TRY(prg_map2(ctx, seed01, G.c(b01), seed01, G.c(b01 + 3 * n_total), n, [] __device__(u64 a, u64 b) { return a + b; },
                      [=] __device__(u64 i, u64 v) { o0[i] = v; })); // A0j[0] + A12j[0] = C01 + C10
    return prg_map1(ctx, seed12, G.c(b12), n, [=] __device__(u64 i, u64 c12) {
        const u64 a0 = (B1[i] == 1 ? f0[n + i] : f0[i]) ^ f2[i]; // :811
        o1[i] = a0 + c12;                                        // A0j[1] + A12j[1]
    });
}
int mpcx_aby3_bit_inject_p2_pre(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                int64_t *to_p0, int64_t *to_p1) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p2_pre");
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && B1 && to_p0 && to_p1, "mpcx_aby3_bit_inject_p2_pre: NULL pointer");
    const BiGeom G{n_total, i0, n};
    MPCX_TRY(bi_select_pairs(ctx, seed20, G.w(PRG_POP_T0 + pop20), n, B0, (u64 *)to_p1)); // W02_b (:888)
    return bi_select_pairs(ctx, seed12, G.w(PRG_POP_T0 + pop12), n, B1, (u64 *)to_p0);    // W12_b (:922)
}
int mpcx_aby3_bit_inject_p2_post(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                 uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p1, int64_t *C0, int64_t *C1) {
    BI_RANGE_CHECK("mpcx_aby3_bit_inject_p2_post");
    if (n == 0) return 0;
    MPCX_REQUIRE(B0 && B1 && from_p0 && from_p1 && C0 && C1, "mpcx_aby3_bit_inject_p2_post: NULL pointer");
    const BiGeom G{n_total, i0, n};
    const u64 *f0 = (const u64 *)from_p0, *f1 = (const u64 *)from_p1;
    u64 *o0 = (u64 *)C0, *o1 = (u64 *)C1;
    MPCX_TRY(prg_map1(ctx, seed12, G.c(PRG_POP_T0 + pop12), n, [=] __device__(u64 i, u64 c12) {
        const u64 a0 = (B0[i] == 1 ? f0[n + i] : f0[i]) ^ f1[i]; // :963
        o0[i] = a0 + c12;                                        // A0j[0] + A12j[0]
    }));
    return prg_map1(ctx, seed20, G.c(PRG_POP_T0 + pop20), n, [=] __device__(u64 i, u64 c02) {
        const u64 a12 = (B1[i] == 1 ? f1[2 * n + i] : f1[n + i]) ^ f0[2 * n + i]; // :996
        o1[i] = c02 + a12;                                                         // A0j[1] + A12j[1]
    });
}

} // extern "C"
