// Per-party local segments of MyEngine (mpc_engine/MyEngine.cpp) as HIP kernels.
//
// MyEngine keeps two-party additive shares on parties 0/1 and uses party 2 as a dealer.  Its correlated
// randomness is read from the *buffers* of the pairwise generators: party 2's prgs[0] / prgs[1] (KEY20 / KEY12,
// MyEngine.cpp:21-24) are the streams parties 0 / 1 hold as prgs[2] (:11-19), so both ends see the same
// int_buff / bool_buff words and only the dealer's correction terms travel.  The buffers are regenerated on the
// device for any index.  share / reveal / privateCompare's helper role are the SecureNN entry points.
#include "mpcx_internal.h"
#include "prg_map.h"

namespace {

constexpr u64 L_1 = ~0ull; // utils/util.hpp:12

// MyEngine.cpp:39-52
__device__ inline u64 add_L_1(u64 a, u64 b) {
    const u64 c = a + b;
    const u64 carry = c < a;
    return (c + carry) % L_1;
}

template <class F> __global__ __launch_bounds__(256) void ew_kernel(u64 n, F f) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) f(i);
}
template <class F> int ew_launch(mpcx_ctx *ctx, u64 n, F f) {
    if (n == 0) return 0;
    u64 blocks = ceil_div(n, 256);
    if (blocks > MPCX_MAX_GRID) blocks = MPCX_MAX_GRID;
    hipLaunchKernelGGL(ew_kernel<F>, dim3((u32)blocks), dim3(256), 0, ctx->stream, n, f);
    MPCX_LAUNCH_CHECK();
    return 0;
}

// scratch from the context's block pool; handed back when the call returns (every use is ordered on the stream)
struct Tmp {
    mpcx_ctx *ctx;
    void *p = nullptr;
    explicit Tmp(mpcx_ctx *c) : ctx(c) {}
    int get(u64 bytes) { return mpcx_malloc(ctx, bytes, &p); }
    ~Tmp() {
        if (p) (void)mpcx_free(ctx, p);
    }
    Tmp(const Tmp &) = delete;
    Tmp &operator=(const Tmp &) = delete;
};

// two values a producing lane hands over per element (prg_buff_map)
struct MyPair {
    u64 a, b;
};

// int_buff[first .. first+count) of a stream (and bool_buff into `bools` when given)
inline int buff(mpcx_ctx *ctx, u64 seed, u64 first, u64 count, u64 *ints, i8 *bools = nullptr) {
    return prg_fill_words(ctx, seed, 0, 2 * first, count, 2, (i64 *)ints, bools);
}

// The dealer's segment of bit_inject (MyEngine.cpp:705-745) in one pass, nothing staged in memory.  For element e of
// the call: (U_j, B_j) = (int_buff, bool_buff)[i0 + e] of the stream shared with party j (:711-712), b = B_0 ^ B_1,
// u = U_0 + U_1 (:717-718); the four first shares V_0, W_0, S_0, T_0 are int_buff[s n_total + i0 + e], s = 0..3, of the
// self stream (:730-739) and the second shares are b u - V_0, (1 ^ b) u - W_0, B_0 B_1 - S_0, B_0 (1 ^ B_1) - T_0.
// A lane owns 32 consecutive elements (64 stream words, one checkpoint run per generator), works in windows of 16
// elements, and hands one output array of a window at a time to the wavefront through LDS, which stores it in memory
// order (128-byte runs).
struct MyDealArgs {
    const PrgCk *ck0, *ck1, *ckS;
    u64 t0, tS[4]; // stream word of element 0 of the call in the pair streams / in the four self-stream sections
    u64 *first[4], *second[4]; // V W S T
    u64 n;
};
__global__ __launch_bounds__(256) void my_bi_deal_kernel(MyDealArgs P) {
    constexpr u32 EPL = 32, WIN = 16, PITCH = WIN + 1;
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 lane0 = (u64)blockIdx.x * 256;
    const u64 e0 = (lane0 + tid) * EPL;
    const bool active = e0 < P.n;
    PrgCursor g0, g1, gs[4];
    g0.open(P.ck0, P.t0 + 2 * e0, active);
    g1.open(P.ck1, P.t0 + 2 * e0, active);
#pragma unroll
    for (u32 q = 0; q < 4; q++) gs[q].open(P.ckS, P.tS[q] + 2 * e0, active);
    u64 *row_tile = tile + tid * PITCH;
    // the wavefront stores the window its lanes have parked in the tile at dst[element]
    auto flush = [&](u64 *__restrict__ dst, u32 win) {
        prg_wave_sync();
#pragma unroll 4
        for (u32 j = 0; j < WIN; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + idx / WIN, col = idx % WIN;
            const u64 e = (lane0 + row) * EPL + win * WIN + col;
            if (e < P.n) dst[e] = tile[row * PITCH + col];
        }
        prg_wave_sync();
    };
#pragma unroll 1
    for (u32 win = 0; win < EPL / WIN; win++) {
        u64 u[WIN];
        u32 b0 = 0, b1 = 0;
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            u[k] = g0.take();
            b0 |= (u32)(g0.take() & 1) << k;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            u[k] += g1.take();
            b1 |= (u32)(g1.take() & 1) << k;
        }
        // the bit that multiplies u (sections 0, 1) or stands alone (sections 2, 3), per element of the window
        const u32 sel[4] = {b0 ^ b1, ~(b0 ^ b1), b0 & b1, b0 & ~b1};
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            // the sections' generators are independent: the fences keep the scheduler from running them side by side
            __builtin_amdgcn_sched_barrier(0);
            u64 y[WIN];
#pragma unroll
            for (u32 k = 0; k < WIN; k++) {
                const u64 x = gs[q].take();
                (void)gs[q].next();
                const u64 bit = (sel[q] >> k) & 1;
                row_tile[k] = x;
                y[k] = (q < 2 ? bit * u[k] : bit) - x;
            }
            flush(P.first[q], win);
#pragma unroll
            for (u32 k = 0; k < WIN; k++) row_tile[k] = y[k];
            flush(P.second[q], win);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

} // namespace

extern "C" {

// ---- Beaver matrix product (MyEngine.cpp:145-219), one round
// rows [row0, row0 + rows) of the dealer's C_0 / C_1 (the row-block form that spreads the product over GPUs)
int mpcx_my_beaver_dealer_matmul_rows(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r, uint64_t l,
                                      uint64_t c, uint64_t row0, uint64_t rows, int64_t *C0, int64_t *C1) {
    MPCX_REQUIRE(C0 && C1, "mpcx_my_beaver_dealer_matmul_rows: NULL pointer");
    MPCX_REQUIRE(row0 <= r && rows <= r - row0, "mpcx_my_beaver_dealer_matmul_rows: rows [%llu, +%llu) outside %llu rows",
                 (unsigned long long)row0, (unsigned long long)rows, (unsigned long long)r);
    const u64 na = r * l, nb = l * c, ma = rows * l, mc = rows * c;
    Tmp t(ctx);
    MPCX_TRY(t.get(8 * (2 * ma + 2 * nb)));
    u64 *As = (u64 *)t.p, *A1 = As + ma, *B0 = A1 + ma, *B1 = B0 + nb;
    // :157-171: A_j then B_j are consecutive slices of stream j's buffer
    MPCX_TRY(buff(ctx, seed0, row0 * l, ma, As));
    MPCX_TRY(buff(ctx, seed1, row0 * l, ma, A1));
    MPCX_TRY(buff(ctx, seed0, na, nb, B0));
    MPCX_TRY(buff(ctx, seed1, na, nb, B1));
    MPCX_TRY(buff(ctx, seedS, row0 * c, mc, (u64 *)C0)); // :173-176
    MPCX_TRY(ew_launch(ctx, ma, [=] __device__(u64 i) { As[i] += A1[i]; }));
    // C_1 = (A_0 + A_1)(B_0 + B_1) - C_0  (:178)
    GemmPart p{(const i64 *)As, (const i64 *)B0, (const i64 *)B1};
    return ring_gemm_parts(ctx, &p, 1, rows, l, c, 0, 0, nullptr, C0, C1, 0, 0);
}
int mpcx_my_beaver_dealer_matmul(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r, uint64_t l,
                                 uint64_t c, int64_t *C0, int64_t *C1) {
    return mpcx_my_beaver_dealer_matmul_rows(ctx, seed0, seed1, seedS, r, l, c, 0, r, C0, C1);
}

int mpcx_my_beaver_party_open_matmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, int layout, uint64_t seed2,
                                     uint64_t r, uint64_t l, uint64_t c, int64_t *ef_out) {
    MPCX_REQUIRE(X && Y && ef_out, "mpcx_my_beaver_party_open_matmul: NULL pointer");
    const u64 na = r * l, nb = l * c;
    Tmp t(ctx);
    MPCX_TRY(t.get(8 * (na + nb)));
    MPCX_TRY(buff(ctx, seed2, 0, na + nb, (u64 *)t.p)); // A | B (:187-198)
    // E = X - A, F = Y - B, row-major (:201-205): the same arithmetic as SecureNN's opening
    return mpcx_snn_beaver_party_open_matmul(ctx, X, Y, layout, (const i64 *)t.p, r, l, c, ef_out);
}

// rows [row0, row0 + rows) of Z (row-major): C, and the E parts of ef_own / ef_peer, hold these rows only (rows x c,
// rows x l followed by the whole l x c F); A and B are read from the stream at their places in the whole product
int mpcx_my_beaver_party_finish_matmul_rows(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                            const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, uint64_t row0,
                                            uint64_t rows, int layoutZ, int64_t *Z) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_my_beaver_party_finish_matmul_rows: party %d", party);
    MPCX_REQUIRE(C && ef_own && ef_peer && Z, "mpcx_my_beaver_party_finish_matmul_rows: NULL pointer");
    MPCX_REQUIRE(row0 <= r && rows <= r - row0, "mpcx_my_beaver_party_finish_matmul_rows: rows [%llu, +%llu) outside %llu rows",
                 (unsigned long long)row0, (unsigned long long)rows, (unsigned long long)r);
    const u64 na = r * l, nb = l * c, ma = rows * l, mc = rows * c;
    Tmp t(ctx);
    MPCX_TRY(t.get(8 * (ma + nb + mc)));
    u64 *triple = (u64 *)t.p; // A rows | B | C rows
    MPCX_TRY(buff(ctx, seed2, row0 * l, ma, triple));
    MPCX_TRY(buff(ctx, seed2, na, nb, triple + ma));
    MPCX_TRY(mpcx_d2d(ctx, triple + ma + nb, C, 8 * mc));
    // Z = C + E*B + A*F + E*F*party (:217)
    return mpcx_snn_beaver_party_finish_matmul(ctx, party, (const i64 *)triple, ef_own, ef_peer, rows, l, c, layoutZ, Z);
}
int mpcx_my_beaver_party_finish_matmul(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                       const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int layoutZ,
                                       int64_t *Z) {
    return mpcx_my_beaver_party_finish_matmul_rows(ctx, party, seed2, C, ef_own, ef_peer, r, l, c, 0, r, layoutZ, Z);
}

// ---- Beaver vector product (:221-280), one round
int mpcx_my_beaver_dealer_vecmul(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t n_total,
                                 uint64_t i0, uint64_t n, int64_t *C0, int64_t *C1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(C0 && C1, "mpcx_my_beaver_dealer_vecmul: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_my_beaver_dealer_vecmul: range outside the vector");
    Tmp t(ctx);
    MPCX_TRY(t.get(32 * n));
    u64 *A0 = (u64 *)t.p, *A1 = A0 + n, *B0 = A1 + n, *B1 = B0 + n;
    MPCX_TRY(buff(ctx, seed0, i0, n, A0)); // :238-239
    MPCX_TRY(buff(ctx, seed1, i0, n, A1));
    MPCX_TRY(buff(ctx, seed0, n_total + i0, n, B0));
    MPCX_TRY(buff(ctx, seed1, n_total + i0, n, B1));
    MPCX_TRY(buff(ctx, seedS, i0, n, (u64 *)C0)); // :244
    const u64 *c0 = (const u64 *)C0;
    return ew_launch(ctx, n, [=] __device__(u64 k) { C1[k] = (i64)((A0[k] + A1[k]) * (B0[k] + B1[k]) - c0[k]); }); // :245
}
int mpcx_my_beaver_party_open_vecmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, uint64_t seed2, uint64_t n_total,
                                     uint64_t i0, uint64_t n, int64_t *ef_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(X && Y && ef_out, "mpcx_my_beaver_party_open_vecmul: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_my_beaver_party_open_vecmul: range outside the vector");
    const u64 first[2] = {i0, n_total + i0}; // A, B (:258-259), consumed while they are regenerated
    return prg_buff_map<2, MyPair>(
        ctx, seed2, first, n, [] __device__(const u64(&w)[2], u32) { return MyPair{w[0], w[1]}; },
        [=] __device__(u64 k, MyPair ab) {
            ef_out[k] = (i64)((u64)X[k] - ab.a); // :262-263
            ef_out[n + k] = (i64)((u64)Y[k] - ab.b);
        });
}
int mpcx_my_beaver_party_finish_vecmul(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                       const int64_t *ef_peer, uint64_t n_total, uint64_t i0, uint64_t n, int64_t *Z) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_my_beaver_party_finish_vecmul: party %d", party);
    MPCX_REQUIRE(C && ef_own && ef_peer && Z, "mpcx_my_beaver_party_finish_vecmul: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_my_beaver_party_finish_vecmul: range outside the vector");
    const u64 pid = (u64)party;
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, MyPair>(
        ctx, seed2, first, n, [] __device__(const u64(&w)[2], u32) { return MyPair{w[0], w[1]}; },
        [=] __device__(u64 k, MyPair ab) {
            const u64 E = (u64)ef_peer[k] + (u64)ef_own[k], F = (u64)ef_peer[n + k] + (u64)ef_own[n + k]; // :276-277
            Z[k] = (i64)((u64)C[k] + E * ab.b + ab.a * F + E * F * pid);                                  // :278
        });
}

// ---- msb (:442-597), three rounds, boolean (XOR) output shares
int mpcx_my_msb_helper_deal(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t i0, uint64_t n,
                            int8_t *xlsb0, uint16_t *xbits0, int8_t *xlsb1, uint16_t *xbits1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(xlsb0 && xbits0 && xlsb1 && xbits1, "mpcx_my_msb_helper_deal: NULL pointer");
    MPCX_REQUIRE(((uintptr_t)xbits0 & 15) == 0 && ((uintptr_t)xbits1 & 15) == 0, "mpcx_my_msb_helper_deal: xbits must be 16-byte aligned");
    const u8 *rnd;
    ctx->rand_tick++;
    MPCX_TRY(prg_buff_mod67_cached(ctx, seedS, 0, 64 * i0, 64 * n, 0, 64 * n, 0, &rnd)); // :495-496
    Tmp t(ctx);
    MPCX_TRY(t.get(24 * n));
    u64 *w0 = (u64 *)t.p, *w1 = w0 + n, *xv = w1 + n;
    MPCX_TRY(buff(ctx, seed0, i0, n, w0)); // :483
    MPCX_TRY(buff(ctx, seed1, i0, n, w1));
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * i0, n, 2, nullptr, xlsb0)); // :490
    MPCX_TRY(ew_launch(ctx, n, [=] __device__(u64 k) {
        const u64 x = add_L_1(w0[k] % L_1, w1[k] % L_1); // :488
        xv[k] = x;
        xlsb1[k] = (i8)((i8)(x & 1) ^ xlsb0[k]); // :491
    }));
    return snn_bit_share_launch(ctx, xv, rnd, n, xbits0, xbits1); // :494-499
}
int mpcx_my_msb_helper_beta(mpcx_ctx *ctx, const int8_t *beta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                            uint64_t seedS, int8_t *b0, int8_t *b1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(beta1_plain && b0 && b1, "mpcx_my_msb_helper_beta: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_my_msb_helper_beta: range outside the vector");
    const u64 first[1] = {n_total + i0};
    return prg_buff_map<1, u64>(
        ctx, seedS, first, n, [] __device__(const u64(&)[1], u32 b) { return (u64)b; }, // :519
        [=] __device__(u64 k, u64 b) {
            b0[k] = (i8)b;
            b1[k] = (i8)(beta1_plain[k] ^ (i8)b); // :520
        });
}
int mpcx_my_msb_party_open(mpcx_ctx *ctx, const int64_t *A, uint64_t seed2, uint64_t i0, uint64_t n, int64_t *r_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(A && r_out, "mpcx_my_msb_party_open: NULL pointer");
    const u64 first[1] = {i0};
    return prg_buff_map<1, u64>(
        ctx, seed2, first, n, [] __device__(const u64(&w)[1], u32) { return w[0] % L_1; }, // :541
        [=] __device__(u64 k, u64 x) {
            const u64 d = 2 * (u64)A[k];   // :446
            const u64 y = add_L_1(d, d);   // :451
            r_out[k] = (i64)add_L_1(x, y); // :548
        });
}
int mpcx_my_msb_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0, uint64_t n,
                                uint64_t seed01, int8_t *beta_out, int64_t *R_out) {
    // beta = boolbuf01[k] (:544), R = r_peer +L r_own (:558): what SecureNN's msb_L_1 feeds privateCompare
    return mpcx_snn_msbl1_party_pc_inputs(ctx, r_own, r_peer, i0, n, seed01, beta_out, R_out);
}
int mpcx_my_msb_party_finish(mpcx_ctx *ctx, int party, const int8_t *beta1_j, const int8_t *beta0, const int8_t *xlsb_j,
                             const int64_t *R, uint64_t n, int8_t *B) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_my_msb_party_finish: party %d", party);
    MPCX_REQUIRE((beta1_j && beta0 && xlsb_j && R && B) || n == 0, "mpcx_my_msb_party_finish: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 k) {
        const i8 b = party == 0 ? beta1_j[k] : (i8)(beta1_j[k] ^ beta0[k]); // :578
        const i8 r0 = (i8)((u64)R[k] % 2);                                 // :579
        B[k] = party == 1 ? (i8)(b ^ xlsb_j[k]) : (i8)(b ^ xlsb_j[k] ^ r0); // :580
    });
}

// ---- bit_inject (:599-772), one round
int mpcx_my_bit_inject_dealer(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t n_total, uint64_t i0,
                              uint64_t n, int64_t *V0, int64_t *W0, int64_t *S0, int64_t *T0, int64_t *V1, int64_t *W1,
                              int64_t *S1, int64_t *T1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(V0 && W0 && S0 && T0 && V1 && W1 && S1 && T1, "mpcx_my_bit_inject_dealer: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_my_bit_inject_dealer: range outside the vector");
    MyDealArgs P;
    MPCX_TRY(mpcx_prg_table(ctx, seed0, 0, 2 * (i0 + n) + PRG_CK, &P.ck0));
    MPCX_TRY(mpcx_prg_table(ctx, seed1, 0, 2 * (i0 + n) + PRG_CK, &P.ck1));
    MPCX_TRY(mpcx_prg_table(ctx, seedS, 0, 2 * (3 * n_total + i0 + n) + PRG_CK, &P.ckS));
    P.t0 = 2 * i0;
    int64_t *firsts[4] = {V0, W0, S0, T0}, *seconds[4] = {V1, W1, S1, T1};
    for (int q = 0; q < 4; q++) {
        P.tS[q] = 2 * ((u64)q * n_total + i0);
        P.first[q] = (u64 *)firsts[q], P.second[q] = (u64 *)seconds[q];
    }
    P.n = n;
    const u64 grid = ceil_div(ceil_div(n, 32), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "mpcx_my_bit_inject_dealer: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(my_bi_deal_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, P);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_my_bit_inject_party_pre(mpcx_ctx *ctx, const int64_t *X, const int8_t *A, uint64_t seed2, uint64_t i0, uint64_t n,
                                 int64_t *y_out, int8_t *c_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(X && A && y_out && c_out, "mpcx_my_bit_inject_party_pre: NULL pointer");
    const u64 first[1] = {i0}; // U = int_buff[i0 + i], B = bool_buff[i0 + i] (:610-611)
    return prg_buff_map<1, MyPair, 8>(
        ctx, seed2, first, n, [] __device__(const u64(&w)[1], u32 b) { return MyPair{w[0], (u64)b}; },
        [=] __device__(u64 i, MyPair ub) {
            y_out[i] = (i64)((u64)X[i] - ub.a);   // :619
            c_out[i] = (i8)((i8)ub.b ^ A[i]);     // :617
        });
}
int mpcx_my_bit_inject_party_post(mpcx_ctx *ctx, int party, uint64_t seed2, uint64_t i0, uint64_t n, const int64_t *y_own,
                                  const int8_t *c_own, const int64_t *y_peer, const int8_t *c_peer, const int64_t *V,
                                  const int64_t *W, const int64_t *S, const int64_t *T, int64_t *Z) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_my_bit_inject_party_post: party %d", party);
    MPCX_REQUIRE(y_own && c_own && y_peer && c_peer && V && W && S && T && Z, "mpcx_my_bit_inject_party_post: NULL pointer");
    const u64 pid = (u64)party;
    const u64 first[1] = {i0}; // B = bool_buff[i0 + i] (:611)
    return prg_buff_map<1, u64>(
        ctx, seed2, first, n, [] __device__(const u64(&)[1], u32 b) { return (u64)b; },
        [=] __device__(u64 i, u64 bj) {
            const u64 y = (u64)y_peer[i] + (u64)y_own[i]; // :634
            const i8 cp = (i8)(c_peer[i] ^ c_own[i]);     // :635
            if (cp == 0)
                Z[i] = (i64)((u64)V[i] + y * (bj - 2 * (u64)S[i])); // :654
            else
                Z[i] = (i64)((u64)W[i] + y * ((pid ^ bj) - 2 * (u64)T[i])); // :658
        });
}

// reveal_vec_bools (:777-807): XOR of two byte-per-boolean shares
int mpcx_xor_bools(mpcx_ctx *ctx, const int8_t *a, const int8_t *b, int8_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && out) || n == 0, "mpcx_xor_bools: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { out[i] = (i8)(a[i] ^ b[i]); });
}

} // extern "C"
