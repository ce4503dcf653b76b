// Per-party local segments of SecurennEngine (mpc_engine/SecurennEngine.cpp) as HIP kernels.
//
// Z_67 bit shares travel as 64 uint16 per element (128 B, LSB first).  The privateCompare kernels
// give each element to 8 adjacent lanes (16 bytes = 8 bit positions per lane), so a wavefront moves
// 8 whole elements per 1 KiB load/store and the running sum over bit positions is a short suffix
// scan across the 8 lanes of a group.  Randomness that the reference reads from PRgenerator::int_buff
// is regenerated on the device from the stream checkpoints.
#include <cstdlib>

#include "mpcx_internal.h"
#include "prg_map.h"

namespace {

constexpr u64 L_1 = ~0ull; // utils/util.hpp:12
constexpr u32 P = 67;      // utils/util.hpp:11

// SecurennEngine.cpp:15-28
__device__ inline u64 add_L_1(u64 a, u64 b) {
    const u64 c = a + b;
    const u64 carry = c < a;
    return (c + carry) % L_1;
}
__device__ inline u64 sub_L_1(u64 a, u64 b) { return add_L_1(a, (L_1 - b) % L_1); }

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

struct u16x8 {
    u32 w[4];
    __device__ u32 get(int j) const { return (w[j >> 1] >> (16 * (j & 1))) & 0xFFFF; }
    __device__ void set(int j, u32 v) { w[j >> 1] = (w[j >> 1] & ~(0xFFFFu << (16 * (j & 1)))) | (v << (16 * (j & 1))); }
};

// x % 67 for x < 2^16 without a wide multiply: floor(x / 67) = (x * ceil(2^22 / 67)) >> 22 exactly while
// x * 30 < 2^22 (30 = 62602 * 67 - 2^22), and the product stays below 2^32; three full-rate 24-bit operations.
// (HIP declares __umul24 with a signed result: convert before shifting.)
__device__ inline u32 mul24(u32 a, u32 b) { return (u32)__umul24(a, b); }
__device__ inline u32 mod67_u16(u32 x) {
    const u32 q = mul24(x, 62602u) >> 22;
    return x - mul24(q, P);
}

// privateCompare, party 0/1 (SecurennEngine.cpp:543-592).  One piece = 8 bit positions of one element.
// S / U: one byte per value, already reduced mod 67 with 0 -> 1 (:523-534).
// variant 1 = MyEngine::privateCompare (MyEngine.cpp:316-353): the same two formulas, chosen by beta == 0 alone
// (any other beta compares against r + 1, which wraps to 0 for r = -1); U is not read.
//
// The reference stores every intermediate in a uint16_t before reducing it mod 67, and inputs are not required to
// be below 67, so the truncations are kept: with xm = x % 67, bit b of r (beta = 0) or of r + 1 (beta = 1):
//   w = b ? ((66 x + party) mod 2^16) % 67 : xm                                            (:561-562, :569-570)
//   c = ((beta = 0 ? (66 x) % 67 + party b + party : 66 party b + x + party) + sum_w) mod 2^16 % 67   (:559, :567)
// where (66 x) % 67 = (67 - xm) % 67 because the reference reduces that product before it is truncated.
// SMALL: every x of the wavefront is below 67 (the protocol's domain): x % 67 = x and (66 x + party) % 67 =
// (party - x) mod 67 need no division, and no sum leaves 16 bits.
template <bool SMALL>
__device__ inline void pc_piece(u32 pid, int mode, u32 q, u32 bits8, const u32 (&xw)[4], uint2 sb, u32 (&dw)[4]) {
    u32 xk[8], xm[8], wv[8];
    u32 tot = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        xk[j] = (j & 1) ? xw[j >> 1] >> 16 : xw[j >> 1] & 0xFFFF;
        u32 w1;
        if (SMALL) {
            xm[j] = xk[j];
            const i32 t = (i32)pid - (i32)xk[j];
            w1 = (u32)(t + ((t >> 31) & (i32)P));
        } else {
            xm[j] = mod67_u16(xk[j]);
            w1 = mod67_u16((mul24(xk[j], P - 1) + pid) & 0xFFFF);
        }
        wv[j] = ((bits8 >> j) & 1) ? w1 : xm[j];
        tot += wv[j];
    }
    if (mode == 2) tot = 0;
    // sum of W over all higher bit positions: suffix scan over the 8 lanes of the element
    u32 incl = tot;
#pragma unroll
    for (u32 dlt = 1; dlt < 8; dlt <<= 1) {
        const u32 t = __shfl_down(incl, dlt, 8);
        if (q + dlt < 8) incl += t;
    }
    u32 run = incl - tot + pid; // W of the lanes above this one, plus the constant term of c
    const u32 add1 = mode == 0 ? pid : (P - 1) * pid; // coefficient of the bit
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        const u32 b = (bits8 >> j) & 1;
        // (66 x) % 67 = (67 - xm) % 67; the unreduced 67 - xm gives the same c
        const u32 pre = mode == 0 ? P - xm[j] : xk[j];
        u32 t = pre + run + mul24(b, add1);
        if (!SMALL) t &= 0xFFFF;
        const u32 ck = mod67_u16(t);
        run += wv[j];
        const u32 sk = ((j < 4 ? sb.x : sb.y) >> (8 * (j & 3))) & 0xFF;
        const u32 dk = mod67_u16(mul24(sk, ck)); // :589-590 (both factors are below 67)
        dw[j >> 1] |= dk << (16 * (j & 1));
    }
}

// One piece (8 bit positions of element c >> 3) from its loads to its store; load_sb() = the 8 s bytes of the piece.
template <class SB>
__device__ inline void pc_party_lane(u32 pid, int variant, bool live, u64 c, const uint4 *__restrict__ xbits,
                                     const i8 *__restrict__ beta, const i64 *__restrict__ r, SB load_sb,
                                     const u64 *__restrict__ U, uint4 *__restrict__ d_out) {
    const u64 i = live ? c >> 3 : 0;
    const u32 q = (u32)(c & 7);
    u32 xw[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
    uint2 sb = make_uint2(0, 0);
    u64 rv = 0;
    int bt = 0;
    if (live) {
        const uint4 xv = ld_stream(xbits + c);
        xw[0] = xv.x, xw[1] = xv.y, xw[2] = xv.z, xw[3] = xv.w;
        sb = load_sb();
        rv = (u64)r[i];
        bt = beta[i];
    }
    const int mode = (bt == 0) ? 0 : ((variant == 1 || (bt == 1 && rv != ~0ull)) ? 1 : 2);
    // the 8 bits of r (mode 0) or r + 1 (mode 1) this lane works on
    const u32 bits8 = (u32)(((mode == 0 ? rv : rv + 1) >> (8 * q)) & 0xFF);
    // any 16-bit x above 66?  (above 127, or below 128 with x + 61 reaching 128)
    u32 big = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) big |= (xw[k] & 0xFF80FF80u) | ((xw[k] + 0x003D003Du) & 0x00800080u);
    if (__any(big != 0))
        pc_piece<false>(pid, mode, q, bits8, xw, sb, dw);
    else
        pc_piece<true>(pid, mode, q, bits8, xw, sb, dw);
    if (__any(mode == 2)) { // beta = 1 and r = -1 (:574-584): c comes from u alone
        uint2 ub = make_uint2(0, 0);
        if (mode == 2 && live) ub = ((const uint2 *)U)[c];
        u32 d2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 uk = ((j < 4 ? ub.x : ub.y) >> (8 * (j & 3))) & 0xFF;
            // :577-578: (1 - party)(1 + u) + (67 - party) u stays below 2^16 for u < 67, so it is 1 + 68 u = 1 + u
            // (party 0) or 66 u = -u (party 1) mod 67; the generated u bytes are in [1, 66].  Position 0 (:582): u, -u.
            u32 ck;
            if (pid == 0)
                ck = (q != 0 || j != 0) ? (uk == P - 1 ? 0 : uk + 1) : uk;
            else
                ck = P - uk;
            const u32 sk = ((j < 4 ? sb.x : sb.y) >> (8 * (j & 3))) & 0xFF;
            d2[j >> 1] |= mod67_u16(mul24(sk, ck)) << (16 * (j & 1));
        }
        if (mode == 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) dw[k] = d2[k];
        }
    }
    if (live) st_stream(d_out + c, dw[0], dw[1], dw[2], dw[3]);
}

// s bytes read from memory (the context's cache of generator bytes, or scratch)
__global__ __launch_bounds__(256) void pc_party_kernel(int party, int variant, const uint4 *__restrict__ xbits, const i8 *__restrict__ beta,
                                                       const i64 *__restrict__ r, const u64 *__restrict__ S,
                                                       const u64 *__restrict__ U, u64 n, uint4 *__restrict__ d_out) {
    const u64 pieces = n * 8;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 base = (u64)blockIdx.x * 256; base < pieces; base += stride) {
        const u64 c = base + threadIdx.x; // pieces is a multiple of 8, so groups of 8 lanes are all-live or all-dead
        const bool live = c < pieces;
        pc_party_lane((u32)party, variant, live, c, xbits, beta, r, [&] { return ((const uint2 *)S)[c]; }, U, d_out);
    }
}

// s bytes made on the spot (:523, :525-529 with int_buff regenerated from the stream's checkpoints): a wavefront takes
// 32 elements at a time.  Its 64 lanes first replay one checkpoint run each -- 32 of the (uint16)int_buff % 67 bytes,
// half an element -- into the wavefront's 2 KiB of LDS, then work through the 32 elements in four rounds of eight
// elements x eight pieces, each lane picking its piece's 8 bytes out of LDS.  One pass over the 265 bytes of an
// element, no s bytes in memory; ck0 = the checkpoint of element 0's first run.
__global__ __launch_bounds__(256) void pc_party_gen_kernel(int party, int variant, const uint4 *__restrict__ xbits,
                                                           const i8 *__restrict__ beta, const i64 *__restrict__ r,
                                                           const PrgCk *__restrict__ ck0, const u64 *__restrict__ U, u64 n,
                                                           uint4 *__restrict__ d_out) {
    static_assert(PRG_CK == 64, "an element's 64 bytes are two checkpoint runs");
    __shared__ u32 sbytes[4][512];
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 *mine = sbytes[wid];
    const u64 groups = (n + 31) / 32;
    for (u64 g = (u64)blockIdx.x * 4 + wid; g < groups; g += (u64)gridDim.x * 4) {
        const u64 e0 = g * 32;
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (e0 + lane / 2 < n) mod67_run(ck0[2 * e0 + lane], 1, w);
        *(uint4 *)(mine + 8 * lane) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(mine + 8 * lane + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        prg_wave_sync();
#pragma unroll 1
        for (u32 sub = 0; sub < 4; sub++) {
            const u64 c = (e0 + 8 * sub) * 8 + lane;
            const bool live = c < n * 8;
            // element 8 sub + lane / 8 of the group, piece lane % 8
            pc_party_lane((u32)party, variant, live, c, xbits, beta, r,
                          [&] { return *(const uint2 *)(mine + 2 * (64 * sub + lane)); }, U, d_out);
        }
        prg_wave_sync();
    }
}

// the elements whose c comes from u (beta = 1 and r = -1; the same test as in pc_party_kernel), in any order.
// A workgroup collects the hits of PC_LIST_CHUNK elements in LDS and reserves their places in the list with one
// global atomic: one per wavefront would queue a quarter of a million additions on one address at 2^24 elements.
constexpr u32 PC_LIST_CHUNK = 4096;
__global__ __launch_bounds__(256) void pc_list_u_kernel(const i8 *__restrict__ beta, const i64 *__restrict__ r, u64 n,
                                                        u32 *__restrict__ list, u32 *__restrict__ listed) {
    __shared__ u32 hits[PC_LIST_CHUNK];
    __shared__ u32 count, base;
    const u32 tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) count = 0;
    __syncthreads();
    for (u32 it = 0; it < PC_LIST_CHUNK / 256; it++) {
        const u64 i = (u64)blockIdx.x * PC_LIST_CHUNK + it * 256 + tid;
        bool need = false;
        if (i < n) {
            const int bt = beta[i];
            need = bt != 0 && !(bt == 1 && (u64)r[i] != ~0ull);
        }
        const u64 m = __ballot(need);
        if (m == 0) continue; // the same for the whole wavefront
        u32 at = 0;
        if (lane == 0) at = atomicAdd(&count, (u32)__popcll(m));
        at = __shfl(at, 0);
        if (need) hits[at + __popcll(m & ((1ull << lane) - 1))] = (u32)i;
    }
    __syncthreads();
    if (tid == 0 && count) base = atomicAdd(listed, count);
    __syncthreads();
    for (u32 k = tid; k < count; k += 256) list[base + k] = hits[k];
}

// privateCompare, party 2 (:655-669): beta'[i] = any_k((d0[i][k] + d1[i][k]) % 67 == 0)
__global__ __launch_bounds__(256) void pc_helper_kernel(const uint4 *__restrict__ d0, const uint4 *__restrict__ d1, u64 n,
                                                        i8 *__restrict__ beta_out) {
    const u64 pieces = n * 8;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 base = (u64)blockIdx.x * 256; base < pieces; base += stride) {
        const u64 c = base + threadIdx.x;
        const bool live = c < pieces;
        bool hit = false;
        if (live) {
            const uint4 a = ld_stream(d0 + c), b = ld_stream(d1 + c);
            const u32 aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u32 s = ((aw[j >> 1] >> (16 * (j & 1))) + (bw[j >> 1] >> (16 * (j & 1)))) & 0xFFFF;
                hit |= (s % P) == 0;
            }
        }
        const u64 m = __ballot(hit); // 8 lanes per element, 8 elements per wavefront
        const u32 lane = threadIdx.x & 63;
        if (live && (lane & 7) == 0) beta_out[c >> 3] = (i8)(((m >> lane) & 0xFF) != 0);
    }
}

// Z_67 sharing of the 64 bits of v[i] (:835-842 / :919-927): rnd = one random byte (already % 67) per bit;
// first[i][k] = rnd, second[i][k] = (bit + 67 - rnd) % 67
__device__ inline void bit_share_piece(u64 val, u32 q, u64 rb, u64 c, uint4 *__restrict__ first, uint4 *__restrict__ second) {
    u16x8 a{}, b{};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 s = (u32)((rb >> (8 * j)) & 0xFF);
        const u32 bit = (u32)((val >> (8 * q + j)) & 1);
        a.set(j, s);
        b.set(j, (bit + P - s) % P);
    }
    st_stream(first + c, a.w[0], a.w[1], a.w[2], a.w[3]);
    st_stream(second + c, b.w[0], b.w[1], b.w[2], b.w[3]);
}
__global__ __launch_bounds__(256) void bit_share_kernel(const u64 *__restrict__ v, const u64 *__restrict__ rnd, u64 n,
                                                        uint4 *__restrict__ first, uint4 *__restrict__ second) {
    const u64 pieces = n * 8;
    for (u64 c = (u64)blockIdx.x * 256 + threadIdx.x; c < pieces; c += (u64)gridDim.x * 256)
        bit_share_piece(v[c >> 3], (u32)(c & 7), rnd[c], c, first, second);
}
// the same with the random bytes made on the spot, as pc_party_gen_kernel makes its s bytes: a wavefront replays the 64
// checkpoint runs of 32 elements into its 2 KiB of LDS, then shares the 32 words in four rounds
__global__ __launch_bounds__(256) void bit_share_gen_kernel(const u64 *__restrict__ v, const PrgCk *__restrict__ ck0, u64 n,
                                                            uint4 *__restrict__ first, uint4 *__restrict__ second) {
    static_assert(PRG_CK == 64, "an element's 64 bytes are two checkpoint runs");
    __shared__ u32 rbytes[4][512];
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 *mine = rbytes[wid];
    const u64 groups = (n + 31) / 32;
    for (u64 g = (u64)blockIdx.x * 4 + wid; g < groups; g += (u64)gridDim.x * 4) {
        const u64 e0 = g * 32;
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (e0 + lane / 2 < n) mod67_run(ck0[2 * e0 + lane], 0, w);
        *(uint4 *)(mine + 8 * lane) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(mine + 8 * lane + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        prg_wave_sync();
#pragma unroll 1
        for (u32 sub = 0; sub < 4; sub++) {
            const u64 c = (e0 + 8 * sub) * 8 + lane;
            if (c < n * 8) bit_share_piece(v[c >> 3], lane & 7, *(const u64 *)(mine + 2 * (64 * sub + lane)), c, first, second);
        }
        prg_wave_sync();
    }
}

// The dealer's triples of vector_mul (SecurennEngine.cpp:191-198) in one pass, nothing staged in memory: for party
// section i (0, 1) element j reads int_buff[3 (i n_total + j) + {0, 1, 2}] = stream words 6 (i n_total + j) + {0, 2, 4}
// of the self stream as A_i, B_i, C_i, and C_1 is replaced by (A_0 + A_1)(B_0 + B_1) - C_0 (:198).  A lane owns 32
// consecutive elements (192 stream words, three checkpoint runs) and one generator per section, works in windows of
// 16 elements whose values stay in registers, and hands one output array of a window at a time to the wavefront
// through LDS, which stores it in memory order (128-byte runs).
struct DealVecArgs {
    const PrgCk *ck;
    u64 t0, t1; // stream word of A_0[first element of the call], A_1[..]
    u64 *out[6]; // A0 B0 C0 A1 B1 C1
    u64 n;
};
__global__ __launch_bounds__(256) void beaver_deal_vec_kernel(DealVecArgs P) {
    constexpr u32 EPL = 32, WIN = 16, PITCH = WIN + 1;
    __shared__ u64 tile[256 * PITCH];
    const u32 tid = threadIdx.x;
    const u64 lane0 = (u64)blockIdx.x * 256;
    const u64 e0 = (lane0 + tid) * EPL;
    const bool active = e0 < P.n;
    PrgCursor g0, g1;
    g0.open(P.ck, P.t0 + 6 * e0, active);
    g1.open(P.ck, P.t1 + 6 * e0, active);
    u64 *row_tile = tile + tid * PITCH;
    auto flush = [&](u64 *__restrict__ dst, u32 win, const u64 (&v)[WIN]) {
#pragma unroll
        for (u32 k = 0; k < WIN; k++) row_tile[k] = v[k];
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
        u64 a[WIN], b[WIN], c[WIN];
#pragma unroll
        for (u32 k = 0; k < WIN; k++) { // the odd stream words are the generator's bool_buff entries
            a[k] = g0.take(), (void)g0.next();
            b[k] = g0.take(), (void)g0.next();
            c[k] = g0.take(), (void)g0.next();
        }
        flush(P.out[0], win, a);
        flush(P.out[1], win, b);
        flush(P.out[2], win, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (u32 k = 0; k < WIN; k++) {
            const u64 a1 = g1.take();
            (void)g1.next();
            const u64 b1 = g1.take();
            for (u32 q = 0; q < 3; q++) (void)g1.next(); // bool, the unused C word of section 1, bool
            c[k] = (a[k] + a1) * (b[k] + b1) - c[k];
            a[k] = a1, b[k] = b1;
        }
        flush(P.out[3], win, a);
        flush(P.out[4], win, b);
        flush(P.out[5], win, c);
        __builtin_amdgcn_sched_barrier(0);
    }
}

inline u32 grid_for(u64 work) {
    u64 b = ceil_div(work, 256);
    if (b > MPCX_MAX_GRID) b = MPCX_MAX_GRID;
    if (b == 0) b = 1;
    return (u32)b;
}

// the (r0, r1) -> R, r1', alpha bookkeeping of shareConvert (:700-710)
struct ScRand {
    u64 R, r0, r1;
};
__device__ inline ScRand sc_rand(u64 w0, u64 w1) {
    ScRand o{w0 + w1, w0, w1};
    if (o.R == 0) {
        o.R += 1;
        o.r1 += 1;
    }
    return o;
}

} // namespace

// Z_67 sharing of the bits of v[0 .. n) with the random bytes (uint16)int_buff[rnd_first + 64 i + k] % 67 of generator
// (seed, 0): from the context's byte cache when it is on, otherwise made inside the sharing kernel (runs that start on a
// checkpoint) or into scratch behind the first `ws_front` bytes of the workspace, which hold v
static int bit_share_with_stream(mpcx_ctx *ctx, const u64 *v_in_ws, u64 ws_front, u64 seed, u64 rnd_first, u64 n, u16 *first,
                                 u16 *second) {
    const bool cached = ctx->rand_cap >= 64 * n;
    if (!cached && (2 * rnd_first) % PRG_CK == 0) {
        const PrgCk *ck;
        MPCX_TRY(mpcx_prg_table(ctx, seed, 0, 2 * (rnd_first + 64 * n) + PRG_CK, &ck));
        u64 grid = ceil_div(ceil_div(n, 32), 4);
        if (grid > MPCX_MAX_GRID) grid = MPCX_MAX_GRID;
        ProfScope ps(ctx, PROF_PRG);
        hipLaunchKernelGGL(bit_share_gen_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, v_in_ws, ck + 2 * rnd_first / PRG_CK,
                           n, (uint4 *)first, (uint4 *)second);
        MPCX_LAUNCH_CHECK();
        return 0;
    }
    const u8 *rnd;
    ctx->rand_tick++;
    // (the workspace was sized for this fallback by the caller, so the request below does not move it)
    MPCX_TRY(prg_buff_mod67_cached(ctx, seed, 0, rnd_first, 64 * n, 0, ws_front + 64 * n, ws_front, &rnd));
    hipLaunchKernelGGL(bit_share_kernel, dim3(grid_for(8 * n)), dim3(256), 0, ctx->stream, v_in_ws, (const u64 *)rnd, n,
                       (uint4 *)first, (uint4 *)second);
    MPCX_LAUNCH_CHECK();
    return 0;
}

int snn_bit_share_launch(mpcx_ctx *ctx, const u64 *v, const u8 *rnd, u64 n, u16 *first, u16 *second) {
    hipLaunchKernelGGL(bit_share_kernel, dim3(grid_for(8 * n)), dim3(256), 0, ctx->stream, v, (const u64 *)rnd, n,
                       (uint4 *)first, (uint4 *)second);
    MPCX_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int mpcx_snn_share_local(mpcx_ctx *ctx, int party, const int64_t *X, uint64_t n, uint64_t seed01, uint64_t pop01,
                         int64_t *out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party >= 0 && party <= 2, "mpcx_snn_share_local: party %d", party);
    MPCX_REQUIRE(out && (party != 1 || X), "mpcx_snn_share_local: NULL pointer");
    if (party == 2) {
        MPCX_HIP(hipMemsetAsync(out, 0, 8 * n, ctx->stream));
        return 0;
    }
    u64 *o = (u64 *)out;
    const u64 *x = (const u64 *)X;
    if (party == 0) return prg_map1(ctx, seed01, PRG_POP_T0 + pop01, n, [=] __device__(u64 i, u64 w) { o[i] = 0 - w; });
    return prg_map1(ctx, seed01, PRG_POP_T0 + pop01, n, [=] __device__(u64 i, u64 w) { o[i] = x[i] + w; });
}

int mpcx_snn_add_L1(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n) {
    MPCX_REQUIRE((a && b && out) || n == 0, "mpcx_snn_add_L1: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) { out[i] = (i64)add_L_1((u64)a[i], (u64)b[i]); });
}
int mpcx_snn_add_zp(mpcx_ctx *ctx, const uint16_t *a, const uint16_t *b, uint16_t *out, uint64_t n64) {
    MPCX_REQUIRE((a && b && out) || n64 == 0, "mpcx_snn_add_zp: NULL pointer");
    return ew_launch(ctx, n64, [=] __device__(u64 i) { out[i] = (u16)(((u32)(a[i] + b[i]) & 0xFFFF) % P); });
}

// ---- Beaver matmul
// rows [row0, row0 + rows) of the dealer's triple of an r x l . l x c product: the rows of A_i, C_i and -- when B0 / B1
// are given -- the whole of B_i (SecurennEngine.cpp:80, :92, :103: consecutive slices of the self stream's buffer),
// C_1 = (A_0 + A_1)(B_0 + B_1) - C_0 (:106-107) for these rows.  B0 / B1 may be NULL for a caller that needs only the
// row parts (the B slices are then made in scratch).
int mpcx_snn_beaver_dealer_matmul_rows(mpcx_ctx *ctx, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, uint64_t row0,
                                       uint64_t rows, int64_t *A0, int64_t *A1, int64_t *B0, int64_t *B1, int64_t *C0,
                                       int64_t *C1) {
    MPCX_REQUIRE(A0 && A1 && C0 && C1, "mpcx_snn_beaver_dealer_matmul_rows: NULL pointer");
    MPCX_REQUIRE(!B0 == !B1, "mpcx_snn_beaver_dealer_matmul_rows: B0 and B1 go together");
    MPCX_REQUIRE(row0 <= r && rows <= r - row0, "mpcx_snn_beaver_dealer_matmul_rows: rows [%llu, +%llu) outside %llu rows",
                 (unsigned long long)row0, (unsigned long long)rows, (unsigned long long)r);
    const u64 na = r * l, nb = l * c, ma = rows * l, mc = rows * c;
    const u64 gemm_bytes = ring_gemm_ws_bytes(1, rows, l, c);
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, gemm_bytes + 8 * (ma + (B0 ? 0 : 2 * nb)), &ws));
    u64 *As = (u64 *)((u8 *)ws + gemm_bytes);
    i64 *b0 = B0 ? B0 : (i64 *)(As + ma), *b1 = B1 ? B1 : b0 + nb;
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * (row0 * l), ma, 2, A0, nullptr));
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * (na + row0 * l), ma, 2, A1, nullptr));
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * (2 * na), nb, 2, b0, nullptr));
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * (2 * na + nb), nb, 2, b1, nullptr));
    MPCX_TRY(prg_fill_words(ctx, seedS, 0, 2 * (2 * na + 2 * nb + row0 * c), mc, 2, C0, nullptr));
    const u64 *a0 = (const u64 *)A0, *a1 = (const u64 *)A1;
    MPCX_TRY(ew_launch(ctx, ma, [=] __device__(u64 i) { As[i] = a0[i] + a1[i]; }));
    GemmPart p{(const i64 *)As, b0, b1};
    return ring_gemm_parts(ctx, &p, 1, rows, l, c, 0, 0, nullptr, C0, C1, 0, 0);
}

int mpcx_snn_beaver_dealer_matmul(mpcx_ctx *ctx, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, int64_t *msg0,
                                  int64_t *msg1) {
    MPCX_REQUIRE(msg0 && msg1, "mpcx_snn_beaver_dealer_matmul: NULL pointer");
    const u64 na = r * l, nb = l * c;
    return mpcx_snn_beaver_dealer_matmul_rows(ctx, seedS, r, l, c, 0, r, msg0, msg1, msg0 + na, msg1 + na, msg0 + na + nb,
                                              msg1 + na + nb);
}

int mpcx_snn_beaver_party_open_matmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, int layout,
                                      const int64_t *triple, uint64_t r, uint64_t l, uint64_t c, int64_t *ef_out) {
    MPCX_REQUIRE(X && Y && triple && ef_out, "mpcx_snn_beaver_party_open_matmul: NULL pointer");
    const u64 na = r * l, nb = l * c;
    const u64 *x = (const u64 *)X, *y = (const u64 *)Y, *t = (const u64 *)triple;
    u64 *o = (u64 *)ef_out;
    return ew_launch(ctx, na + nb, [=] __device__(u64 i) { // :150-155, E | F row-major
        if (i < na) {
            const u64 row = i / l, col = i - row * l;
            o[i] = x[layout ? col * r + row : i] - t[i];
        } else {
            const u64 j = i - na, row = j / c, col = j - row * c;
            o[i] = y[layout ? col * l + row : j] - t[i];
        }
    });
}

int mpcx_snn_beaver_party_finish_matmul(mpcx_ctx *ctx, int party, const int64_t *triple, const int64_t *ef_own,
                                        const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int layoutZ,
                                        int64_t *Z) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_beaver_party_finish_matmul: party %d", party);
    MPCX_REQUIRE(triple && ef_own && ef_peer && Z, "mpcx_snn_beaver_party_finish_matmul: NULL pointer");
    const u64 na = r * l, nb = l * c;
    const u64 gemm_bytes = ring_gemm_ws_bytes(2, r, l, c);
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, gemm_bytes + 8 * (na + nb), &ws));
    u64 *EF = (u64 *)((u8 *)ws + gemm_bytes);
    const u64 *a = (const u64 *)ef_own, *b = (const u64 *)ef_peer;
    MPCX_TRY(ew_launch(ctx, na + nb, [=] __device__(u64 i) { EF[i] = a[i] + b[i]; })); // :165-166
    // Z = C + E*B + A*F + party*E*F (:167) = C + [E | A] * [B + party*F ; F]
    const i64 *A = triple, *B = triple + na, *C = triple + na + nb;
    const i64 *E = (const i64 *)EF, *F = (const i64 *)(EF + na);
    GemmPart parts[2] = {{E, B, party ? F : nullptr}, {A, F, nullptr}};
    return ring_gemm_parts(ctx, parts, 2, r, l, c, 0, 0, C, nullptr, Z, layoutZ, 0);
}

// ---- Beaver vector product
int mpcx_snn_beaver_dealer_vecmul(mpcx_ctx *ctx, uint64_t seedS, uint64_t n_total, uint64_t i0, uint64_t n,
                                  int64_t *A0, int64_t *B0, int64_t *C0, int64_t *A1, int64_t *B1, int64_t *C1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(A0 && B0 && C0 && A1 && B1 && C1, "mpcx_snn_beaver_dealer_vecmul: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_beaver_dealer_vecmul: range outside the vector");
    DealVecArgs P;
    MPCX_TRY(mpcx_prg_table(ctx, seedS, 0, 6 * (n_total + i0 + n) + PRG_CK, &P.ck));
    P.t0 = 6 * i0, P.t1 = 6 * (n_total + i0); // :191-193 with i = 0, 1
    int64_t *outs[6] = {A0, B0, C0, A1, B1, C1};
    for (int k = 0; k < 6; k++) P.out[k] = (u64 *)outs[k];
    P.n = n;
    const u64 grid = ceil_div(ceil_div(n, 32), 256);
    MPCX_REQUIRE(grid < (1ull << 31), "mpcx_snn_beaver_dealer_vecmul: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(beaver_deal_vec_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, P);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_snn_beaver_party_open_vecmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, const int64_t *A,
                                      const int64_t *B, uint64_t n, int64_t *E, int64_t *F) {
    MPCX_REQUIRE((X && Y && A && B && E && F) || n == 0, "mpcx_snn_beaver_party_open_vecmul: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        E[i] = (i64)((u64)X[i] - (u64)A[i]);
        F[i] = (i64)((u64)Y[i] - (u64)B[i]);
    });
}
int mpcx_snn_beaver_party_finish_vecmul(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *B,
                                        const int64_t *C, const int64_t *E_own, const int64_t *F_own,
                                        const int64_t *E_peer, const int64_t *F_peer, uint64_t n, int64_t *Z) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_beaver_party_finish_vecmul: party %d", party);
    MPCX_REQUIRE((A && B && C && E_own && F_own && E_peer && F_peer && Z) || n == 0,
                 "mpcx_snn_beaver_party_finish_vecmul: NULL pointer");
    const u64 pid = (u64)party;
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 E = (u64)E_own[i] + (u64)E_peer[i], F = (u64)F_own[i] + (u64)F_peer[i];
        Z[i] = (i64)((u64)C[i] + E * (u64)B[i] + F * (u64)A[i] + F * E * pid); // :294
    });
}

// ---- privateCompare
int mpcx_snn_pc_party(mpcx_ctx *ctx, int party, const uint16_t *xbits, const int8_t *beta, const int64_t *r,
                      uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01, uint16_t *d_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_pc_party: party %d", party);
    MPCX_REQUIRE(xbits && beta && r && d_out, "mpcx_snn_pc_party: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_pc_party: range outside the vector");
    MPCX_REQUIRE(((uintptr_t)xbits & 15) == 0 && ((uintptr_t)d_out & 15) == 0, "mpcx_snn_pc_party: xbits/d_out must be 16-byte aligned");
    MPCX_REQUIRE(n < (1ull << 32), "mpcx_snn_pc_party: too many elements for one call");
    const u8 *S = nullptr, *U;
    ctx->rand_tick++;
    // With the byte cache on (the counterpart of the reference's constructor-filled int_buff) s and u are read from it.
    // Without it the s bytes are made inside the comparison kernel and never stored; the u bytes, which only the
    // elements with beta = 1 and r = -1 read, are made for those elements alone (a generator step costs as much as the
    // comparison itself), into scratch: u, then the list of those elements and its length.
    const bool cached = ctx->rand_cap >= 128 * n;
    const u64 ws_list = 64 * n, ws_total = ws_list + 4 * n + 16;
    if (cached) {
        // (a slice that does not fit beside what the cache already holds for this call is made in scratch instead)
        MPCX_TRY(prg_buff_mod67_cached(ctx, seed01, 0, 64 * i0, 64 * n, 1, 128 * n, 0, &S));                      // :523,525-529
        MPCX_TRY(prg_buff_mod67_cached(ctx, seed01, 0, 64 * n_total + 64 * i0, 64 * n, 1, 128 * n, 64 * n, &U));  // :524,530-534
    } else {
        void *ws;
        MPCX_TRY(mpcx_ws(ctx, ws_total, &ws));
        U = (const u8 *)ws;
        u32 *list = (u32 *)((u8 *)ws + ws_list), *listed = list + n;
        MPCX_HIP(hipMemsetAsync(listed, 0, sizeof(u32), ctx->stream));
        hipLaunchKernelGGL(pc_list_u_kernel, dim3((u32)ceil_div(n, PC_LIST_CHUNK)), dim3(256), 0, ctx->stream, beta, r, n, list,
                           listed);
        MPCX_LAUNCH_CHECK();
        MPCX_TRY(prg_fill_buff_mod67_listed(ctx, seed01, 0, 64 * n_total + 64 * i0, n, 1, list, listed, (u8 *)U));
        const PrgCk *ck;
        MPCX_TRY(mpcx_prg_table(ctx, seed01, 0, 2 * 64 * (i0 + n) + PRG_CK, &ck));
        ProfScope ps(ctx, PROF_PC_PARTY);
        u64 grid = ceil_div(ceil_div(n, 32), 4);
        if (grid > MPCX_MAX_GRID) grid = MPCX_MAX_GRID;
        hipLaunchKernelGGL(pc_party_gen_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, party, 0, (const uint4 *)xbits, beta,
                           r, ck + 2 * i0, (const u64 *)U, n, (uint4 *)d_out);
        MPCX_LAUNCH_CHECK();
        return 0;
    }
    ProfScope ps(ctx, PROF_PC_PARTY);
    hipLaunchKernelGGL(pc_party_kernel, dim3(grid_for(8 * n)), dim3(256), 0, ctx->stream, party, 0, (const uint4 *)xbits, beta,
                       r, (const u64 *)S, (const u64 *)U, n, (uint4 *)d_out);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_my_pc_party(mpcx_ctx *ctx, int party, const uint16_t *xbits, const int8_t *beta, const int64_t *r, uint64_t i0,
                     uint64_t n, uint64_t seed01, uint16_t *d_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_my_pc_party: party %d", party);
    MPCX_REQUIRE(xbits && beta && r && d_out, "mpcx_my_pc_party: NULL pointer");
    MPCX_REQUIRE(((uintptr_t)xbits & 15) == 0 && ((uintptr_t)d_out & 15) == 0, "mpcx_my_pc_party: xbits/d_out must be 16-byte aligned");
    const u8 *S;
    ctx->rand_tick++;
    MPCX_TRY(prg_buff_mod67_cached(ctx, seed01, 0, 64 * i0, 64 * n, 1, 64 * n, 0, &S)); // MyEngine.cpp:298-308
    ProfScope ps(ctx, PROF_PC_PARTY);
    hipLaunchKernelGGL(pc_party_kernel, dim3(grid_for(8 * n)), dim3(256), 0, ctx->stream, party, 1, (const uint4 *)xbits, beta,
                       r, (const u64 *)S, (const u64 *)nullptr, n, (uint4 *)d_out);
    MPCX_LAUNCH_CHECK();
    return 0;
}
int mpcx_snn_pc_helper(mpcx_ctx *ctx, const uint16_t *d0, const uint16_t *d1, uint64_t n, int8_t *beta_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(d0 && d1 && beta_out, "mpcx_snn_pc_helper: NULL pointer");
    MPCX_REQUIRE(((uintptr_t)d0 & 15) == 0 && ((uintptr_t)d1 & 15) == 0, "mpcx_snn_pc_helper: d0/d1 must be 16-byte aligned");
    ProfScope ps(ctx, PROF_PC_HELPER);
    hipLaunchKernelGGL(pc_helper_kernel, dim3(grid_for(8 * n)), dim3(256), 0, ctx->stream, (const uint4 *)d0,
                       (const uint4 *)d1, n, beta_out);
    MPCX_LAUNCH_CHECK();
    return 0;
}

// ---- shareConvert
// r0 = int_buff01[i0 + i], r1 = int_buff01[n_total + i0 + i], eta'' = bool_buff01[i0 + i] (:700-702): consumed while
// they are regenerated
struct ScPick { // what a party keeps of (r0, r1, eta'')
    u64 rj;     // its own summand, after the r == 0 adjustment
    u64 bits;   // bit 0: alpha = (r < r0) (:710), bit 1: eta''
};
int mpcx_snn_sc_party_open(mpcx_ctx *ctx, int party, const int64_t *A, uint64_t n_total, uint64_t i0, uint64_t n,
                           uint64_t seed01, int64_t *x_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_sc_party_open: party %d", party);
    MPCX_REQUIRE(A && x_out, "mpcx_snn_sc_party_open: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_sc_party_open: range outside the vector");
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, u64>(
        ctx, seed01, first, n,
        [=] __device__(const u64(&w)[2], u32) {
            const ScRand s = sc_rand(w[0], w[1]);
            return party == 0 ? s.r0 : s.r1;
        },
        [=] __device__(u64 i, u64 rj) { x_out[i] = (i64)((u64)A[i] + rj); }); // :718
}
int mpcx_snn_sc_party_pc_inputs(mpcx_ctx *ctx, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                                int8_t *eta2_out, int64_t *rm1_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(eta2_out && rm1_out, "mpcx_snn_sc_party_pc_inputs: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_sc_party_pc_inputs: range outside the vector");
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, ScPick>(
        ctx, seed01, first, n,
        [=] __device__(const u64(&w)[2], u32 eta2) { return ScPick{sc_rand(w[0], w[1]).R - 1, (u64)eta2}; }, // :738
        [=] __device__(u64 i, ScPick v) {
            eta2_out[i] = (i8)v.bits;
            rm1_out[i] = (i64)v.rj;
        });
}
int mpcx_snn_sc_party_finish(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *delta_j,
                             const int64_t *eta1_j, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                             int64_t *y_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_sc_party_finish: party %d", party);
    MPCX_REQUIRE(A && delta_j && eta1_j && y_out, "mpcx_snn_sc_party_finish: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_sc_party_finish: range outside the vector");
    const u64 pid = (u64)party;
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, ScPick>(
        ctx, seed01, first, n,
        [=] __device__(const u64(&w)[2], u32 eta2) {
            const ScRand s = sc_rand(w[0], w[1]);
            return ScPick{party == 0 ? s.r0 : s.r1, (u64)(s.R < s.r0) | (u64)eta2 << 1}; // :710
        },
        [=] __device__(u64 i, ScPick v) {
            const u64 rj = v.rj, alpha = v.bits & 1, eta2 = v.bits >> 1;
            const u64 a = (u64)A[i], x = a + rj;       // :718
            const u64 beta = x < rj;                   // :719
            u64 eta = add_L_1((u64)eta1_j[i], pid * eta2);           // :750
            u64 t = add_L_1((u64)eta1_j[i], (u64)eta1_j[i]) * eta2;  // :751-752
            eta = sub_L_1(eta, t);                                   // :753
            const u64 tj = (1 - pid) * (L_1 - 1 - alpha);            // :758
            u64 theta = add_L_1(beta, tj);                           // :759
            theta = add_L_1(theta, eta);
            theta = add_L_1(theta, (u64)delta_j[i]);
            y_out[i] = (i64)sub_L_1(a, theta); // :765
        });
}
int mpcx_snn_sc_helper_split(mpcx_ctx *ctx, const int64_t *x0, const int64_t *x1, uint64_t n_total, uint64_t i0,
                             uint64_t n, uint64_t seedS, uint16_t *xbits0, int64_t *delta0, uint16_t *xbits1,
                             int64_t *delta1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(x0 && x1 && xbits0 && delta0 && xbits1 && delta1, "mpcx_snn_sc_helper_split: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_sc_helper_split: range outside the vector");
    MPCX_REQUIRE(((uintptr_t)xbits0 & 15) == 0 && ((uintptr_t)xbits1 & 15) == 0, "mpcx_snn_sc_helper_split: xbits must be 16-byte aligned");
    void *ws; // x, then room for the random bytes should they have to be made in scratch
    MPCX_TRY(mpcx_ws(ctx, 72 * n, &ws));
    u64 *x = (u64 *)ws;
    const u64 *a = (const u64 *)x0, *b = (const u64 *)x1;
    const u64 first[1] = {i0};
    MPCX_TRY((prg_buff_map<1, u64>(
        ctx, seedS, first, n, [] __device__(const u64(&w)[1], u32) { return w[0] % L_1; }, // :848-849
        [=] __device__(u64 i, u64 D0) {
            const u64 xv = a[i] + b[i];  // :824
            const u64 delta = xv < a[i]; // :825
            x[i] = xv;
            delta0[i] = (i64)D0;
            delta1[i] = (i64)sub_L_1(delta, D0); // :851
        })));
    return bit_share_with_stream(ctx, x, 8 * n, seedS, 64 * i0, n, xbits0, xbits1); // :838-839
}
int mpcx_snn_sc_helper_eta(mpcx_ctx *ctx, const int8_t *eta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                           uint64_t seedS, int64_t *eta0, int64_t *eta1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(eta1_plain && eta0 && eta1, "mpcx_snn_sc_helper_eta: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_sc_helper_eta: range outside the vector");
    const u64 first[1] = {n_total + i0};
    return prg_buff_map<1, u64>(
        ctx, seedS, first, n, [] __device__(const u64(&w)[1], u32) { return w[0] % L_1; }, // :868-869
        [=] __device__(u64 i, u64 e0) {
            eta0[i] = (i64)e0;
            eta1[i] = (i64)sub_L_1((u64)(i64)eta1_plain[i], e0); // :871
        });
}

// ---- msb_L_1
int mpcx_snn_msbl1_helper_deal(mpcx_ctx *ctx, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seedS, int64_t *x0,
                               int64_t *xlsb0, uint16_t *xbits0, int64_t *x1, int64_t *xlsb1, uint16_t *xbits1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(x0 && xlsb0 && xbits0 && x1 && xlsb1 && xbits1, "mpcx_snn_msbl1_helper_deal: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_msbl1_helper_deal: range outside the vector");
    MPCX_REQUIRE(((uintptr_t)xbits0 & 15) == 0 && ((uintptr_t)xbits1 & 15) == 0, "mpcx_snn_msbl1_helper_deal: xbits must be 16-byte aligned");
    void *ws; // x, then room for the random bytes should they have to be made in scratch
    MPCX_TRY(mpcx_ws(ctx, 72 * n, &ws));
    u64 *xv = (u64 *)ws;
    struct Deal {
        u64 X0, X1, lsb0;
    };
    const u64 first[3] = {i0, n_total + i0, 2 * n_total + i0};
    MPCX_TRY((prg_buff_map<3, Deal, 16>(
        ctx, seedS, first, n,
        [] __device__(const u64(&w)[3], u32) { return Deal{w[0] % L_1, w[1] % L_1, w[2]}; }, // :911-914, :917
        [=] __device__(u64 i, Deal d) {
            const u64 X = add_L_1(d.X0, d.X1); // :915
            xv[i] = X;
            x0[i] = (i64)d.X0;
            x1[i] = (i64)d.X1;
            xlsb0[i] = (i64)d.lsb0;
            xlsb1[i] = (i64)((X & 1) - d.lsb0); // :918
        })));
    // :922-926: the random byte is party 1's share, party 0 gets bit + 67 - it
    return bit_share_with_stream(ctx, xv, 8 * n, seedS, 3 * n_total + 64 * i0, n, xbits1, xbits0);
}
int mpcx_snn_msbl1_helper_beta(mpcx_ctx *ctx, const int8_t *beta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                               uint64_t seedS, int64_t *b0, int64_t *b1) {
    if (n == 0) return 0;
    MPCX_REQUIRE(beta1_plain && b0 && b1, "mpcx_snn_msbl1_helper_beta: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_msbl1_helper_beta: range outside the vector");
    const u64 first[1] = {4 * n_total + i0};
    return prg_buff_map<1, u64>(
        ctx, seedS, first, n, [] __device__(const u64(&w)[1], u32) { return w[0]; }, // :953
        [=] __device__(u64 i, u64 w) {
            b0[i] = (i64)w;
            b1[i] = (i64)((u64)(i64)beta1_plain[i] - w); // :954
        });
}
int mpcx_snn_msbl1_party_open(mpcx_ctx *ctx, const int64_t *A, const int64_t *xj, uint64_t n, int64_t *r_out) {
    MPCX_REQUIRE((A && xj && r_out) || n == 0, "mpcx_snn_msbl1_party_open: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 y = add_L_1((u64)A[i], (u64)A[i]); // :1004
        r_out[i] = (i64)add_L_1(y, (u64)xj[i]);      // :1005
    });
}
int mpcx_snn_msbl1_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0,
                                   uint64_t n, uint64_t seed01, int8_t *beta_out, int64_t *R_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(r_own && r_peer && beta_out && R_out, "mpcx_snn_msbl1_party_pc_inputs: NULL pointer");
    MPCX_TRY(prg_fill_words(ctx, seed01, 0, 2 * i0, n, 2, nullptr, beta_out)); // :985
    return ew_launch(ctx, n, [=] __device__(u64 i) { R_out[i] = (i64)add_L_1((u64)r_peer[i], (u64)r_own[i]); }); // :1013
}
int mpcx_snn_msbl1_party_mid(mpcx_ctx *ctx, int party, const int64_t *beta1_j, const int64_t *xlsb_j,
                             const int8_t *beta0, const int64_t *R, uint64_t n, int64_t *gamma_out,
                             int64_t *delta_out) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_msbl1_party_mid: party %d", party);
    MPCX_REQUIRE((beta1_j && xlsb_j && beta0 && R && gamma_out && delta_out) || n == 0,
                 "mpcx_snn_msbl1_party_mid: NULL pointer");
    const u64 pid = (u64)party;
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 r0 = (u64)R[i] % 2, b0 = (u64)(i64)beta0[i]; // :1025
        gamma_out[i] = (i64)((u64)beta1_j[i] + pid * b0 - 2 * b0 * (u64)beta1_j[i]); // :1030
        delta_out[i] = (i64)((u64)xlsb_j[i] + pid * r0 - 2 * r0 * (u64)xlsb_j[i]);   // :1031
    });
}
int mpcx_snn_msbl1_party_finish(mpcx_ctx *ctx, const int64_t *gamma, const int64_t *delta, const int64_t *theta,
                                uint64_t n, int64_t *B) {
    MPCX_REQUIRE((gamma && delta && theta && B) || n == 0, "mpcx_snn_msbl1_party_finish: NULL pointer");
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        B[i] = (i64)((u64)gamma[i] + (u64)delta[i] - 2 * (u64)theta[i]); // :1036
    });
}

// ---- msb (:1089-1101 = shareConvert of 2a, then msb_L_1): the local segments of parties 0 / 1 that fall into one gap
// between two messenger rounds, as one call each.  A party that sends nothing in a round (shareConvert's round 4 into
// msb_L_1's round 1; msb_L_1's round 4 into vector_mul's round 1) may do the work on either side of it, so the work is
// done once both inputs are there.
// before shareConvert's round 1 (:1094-1098, :718): x_j = 2 a_j + r_j
int mpcx_snn_msb_party_open(mpcx_ctx *ctx, int party, const int64_t *A, uint64_t n_total, uint64_t i0, uint64_t n,
                            uint64_t seed01, int64_t *x_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_msb_party_open: party %d", party);
    MPCX_REQUIRE(A && x_out, "mpcx_snn_msb_party_open: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_msb_party_open: range outside the vector");
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, u64>(
        ctx, seed01, first, n,
        [=] __device__(const u64(&w)[2], u32) {
            const ScRand s = sc_rand(w[0], w[1]);
            return party == 0 ? s.r0 : s.r1;
        },
        [=] __device__(u64 i, u64 rj) { x_out[i] = (i64)(((u64)A[i] << 1) + rj); });
}
// after msb_L_1's round 1: the end of shareConvert (:750-765, on the 2 a_j of :1094-1098) and r_j = 2 y_j + x_j over
// Z_{2^64-1} (:1004-1005)
int mpcx_snn_msb_party_convert_open(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *delta_j,
                                    const int64_t *eta1_j, const int64_t *xj, uint64_t n_total, uint64_t i0, uint64_t n,
                                    uint64_t seed01, int64_t *r_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_msb_party_convert_open: party %d", party);
    MPCX_REQUIRE(A && delta_j && eta1_j && xj && r_out, "mpcx_snn_msb_party_convert_open: NULL pointer");
    MPCX_REQUIRE(i0 + n <= n_total, "mpcx_snn_msb_party_convert_open: range outside the vector");
    const u64 pid = (u64)party;
    const u64 first[2] = {i0, n_total + i0};
    return prg_buff_map<2, ScPick>(
        ctx, seed01, first, n,
        [=] __device__(const u64(&w)[2], u32 eta2) {
            const ScRand s = sc_rand(w[0], w[1]);
            return ScPick{party == 0 ? s.r0 : s.r1, (u64)(s.R < s.r0) | (u64)eta2 << 1};
        },
        [=] __device__(u64 i, ScPick v) {
            const u64 rj = v.rj, alpha = v.bits & 1, eta2 = v.bits >> 1;
            const u64 a = (u64)A[i] << 1, x = a + rj;
            const u64 beta = x < rj;
            u64 eta = add_L_1((u64)eta1_j[i], pid * eta2);
            const u64 t = add_L_1((u64)eta1_j[i], (u64)eta1_j[i]) * eta2;
            eta = sub_L_1(eta, t);
            u64 theta = add_L_1(beta, (1 - pid) * (L_1 - 1 - alpha));
            theta = add_L_1(theta, eta);
            theta = add_L_1(theta, (u64)delta_j[i]);
            const u64 y = sub_L_1(a, theta);                          // :765
            r_out[i] = (i64)add_L_1(add_L_1(y, y), (u64)xj[i]);       // :1004-1005
        });
}
// after msb_L_1's round 2 (:985, :1013): beta = bool_buff01[i0 + i], R = r_peer +L r_own, in one pass
int mpcx_snn_msb_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0, uint64_t n,
                                 uint64_t seed01, int8_t *beta_out, int64_t *R_out) {
    if (n == 0) return 0;
    MPCX_REQUIRE(r_own && r_peer && beta_out && R_out, "mpcx_snn_msb_party_pc_inputs: NULL pointer");
    const u64 first[1] = {i0};
    return prg_buff_map<1, u64>(
        ctx, seed01, first, n, [] __device__(const u64(&)[1], u32 b) { return (u64)b; },
        [=] __device__(u64 i, u64 b) {
            beta_out[i] = (i8)b;
            R_out[i] = (i64)add_L_1((u64)r_peer[i], (u64)r_own[i]);
        });
}
// after vector_mul's round 1 (the triple has arrived): gamma_j, delta_j of :1025-1031 and the masked operands
// E = gamma - A, F = delta - B of vector_mul (:253-262)
int mpcx_snn_msb_party_mid_open(mpcx_ctx *ctx, int party, const int64_t *beta1_j, const int64_t *xlsb_j,
                                const int8_t *beta0, const int64_t *R, const int64_t *A, const int64_t *B, uint64_t n,
                                int64_t *gamma_out, int64_t *delta_out, int64_t *E, int64_t *F) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_msb_party_mid_open: party %d", party);
    MPCX_REQUIRE((beta1_j && xlsb_j && beta0 && R && A && B && E && F) || n == 0, "mpcx_snn_msb_party_mid_open: NULL pointer");
    MPCX_REQUIRE(!gamma_out == !delta_out, "mpcx_snn_msb_party_mid_open: gamma_out and delta_out go together");
    const u64 pid = (u64)party;
    const bool keep = gamma_out != nullptr; // without them mpcx_snn_msb_party_finish forms gamma = E + A, delta = F + B again
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 r0 = (u64)R[i] % 2, b0 = (u64)(i64)beta0[i];
        const u64 g = (u64)beta1_j[i] + pid * b0 - 2 * b0 * (u64)beta1_j[i];
        const u64 d = (u64)xlsb_j[i] + pid * r0 - 2 * r0 * (u64)xlsb_j[i];
        if (keep) {
            gamma_out[i] = (i64)g;
            delta_out[i] = (i64)d;
        }
        E[i] = (i64)(g - (u64)A[i]);
        F[i] = (i64)(d - (u64)B[i]);
    });
}
// after vector_mul's round 2: theta_j (:294) and b_j = gamma_j + delta_j - 2 theta_j (:1036)
int mpcx_snn_msb_party_finish(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *B, const int64_t *C,
                              const int64_t *E_own, const int64_t *F_own, const int64_t *E_peer, const int64_t *F_peer,
                              const int64_t *gamma, const int64_t *delta, uint64_t n, int64_t *Bout) {
    MPCX_REQUIRE(party == 0 || party == 1, "mpcx_snn_msb_party_finish: party %d", party);
    MPCX_REQUIRE((A && B && C && E_own && F_own && E_peer && F_peer && Bout) || n == 0, "mpcx_snn_msb_party_finish: NULL pointer");
    MPCX_REQUIRE(!gamma == !delta, "mpcx_snn_msb_party_finish: gamma and delta go together");
    const u64 pid = (u64)party;
    const bool given = gamma != nullptr;
    return ew_launch(ctx, n, [=] __device__(u64 i) {
        const u64 a = (u64)A[i], b = (u64)B[i], eo = (u64)E_own[i], fo = (u64)F_own[i];
        const u64 Ev = eo + (u64)E_peer[i], Fv = fo + (u64)F_peer[i];
        const u64 theta = (u64)C[i] + Ev * b + Fv * a + Fv * Ev * pid;
        // gamma + delta: read, or formed again from the party's own masked operands E = gamma - A, F = delta - B
        const u64 gd = given ? (u64)gamma[i] + (u64)delta[i] : eo + a + fo + b;
        Bout[i] = (i64)(gd - 2 * theta);
    });
}

} // extern "C"
