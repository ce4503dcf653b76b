// Exact Z_2^64 matrix product through a residue number system on the int8 matrix cores of gfx950.
//
// A ring element is read as the integer v = sum d_j 256^j of its eight signed base-256 digits (|v| < 2^63.01,
// v = x mod 2^64).  The integer product X = sum_k v(a_k) v(b_k) has |X| <= K * 2^126.023.  For the NP = 18 largest
// pairwise coprime moduli m_i <= 256 (M = prod m_i = 2^140.45):
//     split   : plane_i = a signed byte congruent to v mod m_i, |.| <= 128            (one pass over the operands)
//     gemm    : acc_i = A_i * B_i^T in int32 (|acc_i| <= 2^14 K < 2^28), y_i = (acc_i mod m_i) * (M/m_i)^-1 mod m_i
//               -- NP plain int8 products instead of the 36 digit-pair products of the positional form
//     combine : X = sum_i y_i (M/m_i) - q M with q = round(sum_i y_i / m_i): for K <= 8192 |X| / M <= 0.369, so the
//               fraction of the float sum stays 0.13 away from one half, five orders above its rounding error;
//               everything is then taken mod 2^64.
// K is limited to 8192 per accumulation chunk (all concatenated parts), as in the positional kernel.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "mpcx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int NP = 18;
// 256, 255 = 3*5*17, 253 = 11*23, 247 = 13*19, 217 = 7*31 and primes: no common factors
constexpr u32 RNS_MODULI[NP] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181};
constexpr u32 TM = 256, TN = 256, RK = 32, SKB = 3; // block tile; 32 bytes of K per block, three K blocks per stage
constexpr u32 KB_BYTES = (TM + TN) * RK;             // 16 KiB: one K block of the A and the B tile
constexpr u32 STAGE_BYTES = SKB * KB_BYTES;          // 48 KiB

struct RnsTab {
    u32 m[NP];       // modulus
    u64 magic40[NP]; // ceil(2^40 / m): exact quotient of any 32-bit x as (x * magic) >> 40
    i32 clo[NP], chi[NP]; // 256^j mod m (symmetric) for j = 0..3 / 4..7, one signed byte each
    u32 off_acc[NP];      // a multiple of m above 2^28
    u32 inv[NP];          // (M / m)^-1 mod m
    u64 cM[NP];           // (M / m) mod 2^64
    float w[NP];          // 1 / m
    u64 Mmod;             // M mod 2^64
};

const RnsTab &rns_tab() {
    static RnsTab T;
    static std::once_flag once;
    std::call_once(once, [] {
        T.Mmod = 1;
        for (int i = 0; i < NP; i++) T.Mmod *= RNS_MODULI[i];
        for (int i = 0; i < NP; i++) {
            const u32 m = RNS_MODULI[i];
            T.m[i] = m;
            T.magic40[i] = ((1ull << 40) + m - 1) / m;
            u32 pw = 1;
            u32 lo = 0, hi = 0;
            for (int j = 0; j < 8; j++) {
                i32 c = (i32)pw;
                if (c > (i32)(m / 2)) c -= (i32)m;
                if (j < 4)
                    lo |= ((u32)c & 0xFFu) << (8 * j);
                else
                    hi |= ((u32)c & 0xFFu) << (8 * (j - 4));
                pw = (pw * 256u) % m;
            }
            T.clo[i] = (i32)lo;
            T.chi[i] = (i32)hi;
            T.off_acc[i] = m * ((1u << 28) / m + 1);
            u64 c64 = 1;
            u32 cm = 1;
            for (int j = 0; j < NP; j++)
                if (j != i) {
                    c64 *= RNS_MODULI[j];
                    cm = (u32)(((u64)cm * (RNS_MODULI[j] % m)) % m);
                }
            T.cM[i] = c64;
            u32 inv = 1;
            while (((u64)inv * cm) % m != 1) inv++;
            T.inv[i] = inv;
            T.w[i] = 1.0f / (float)m;
        }
    });
    return T;
}

struct SplitJob {
    const u64 *src, *src2; // src2 (nullable) is added element-wise (mod 2^64) first
    i8 *planes;
    u64 O, so, sk, Op, koff, kcover; // kcover: K extent this job writes (zeros beyond its K)
};
struct SplitJobs {
    SplitJob j[16];
};

// planes[i][kb][o][k%32] = v(src(o,k) (+ src2(o,k))) mod m_i, K-blocked like the positional planes; zeros outside
// [0,O) x [0,K).  grid (kcover_max/64, max Op/64, jobs).
template <bool KFAST>
__global__ __launch_bounds__(256) void rns_split_kernel(SplitJobs jobs, u64 K, u64 Kp, RnsTab T) {
    const SplitJob &J = jobs.j[blockIdx.z];
    const u32 tid = threadIdx.x;
    const u64 o0 = (u64)blockIdx.y * 64, k0 = (u64)blockIdx.x * 64;
    if (o0 >= J.Op || k0 >= J.kcover) return;
    const u64 *__restrict__ src = J.src, *__restrict__ src2 = J.src2;
    const u64 O = J.O, so = J.so, sk = J.sk;
    const u32 oo = ((tid >> 1) & 31) | ((tid >> 7) << 5), kc = (tid & 1) | ((tid >> 6) & 1) << 1;
    u64 y[16];
    if constexpr (KFAST) {
        __shared__ u64 tile[64][65];
        u64 v[16];
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            const u64 o = o0 + (tid >> 6) + 4 * j, k = k0 + (tid & 63);
            const bool in = o < O && k < K;
            v[j] = in ? src[o * so + k] : 0;
            if (src2 && in) v[j] += src2[o * so + k];
        }
#pragma unroll
        for (u32 j = 0; j < 16; j++) tile[(tid >> 6) + 4 * j][tid & 63] = v[j];
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < 16; q++) y[q] = tile[oo][kc * 16 + q];
    } else {
        const u64 o = o0 + oo;
#pragma unroll
        for (u32 q = 0; q < 16; q++) {
            const u64 k = k0 + kc * 16 + q;
            const bool in = o < O && k < K;
            y[q] = in ? src[o * so + k * sk] : 0;
            if (src2 && in) y[q] += src2[o * so + k * sk];
        }
    }
    // the bytes of (x + 0x80..80) ^ 0x80..80 are the signed digits of x
#pragma unroll
    for (u32 q = 0; q < 16; q++) y[q] = (y[q] + 0x8080808080808080ull) ^ 0x8080808080808080ull;
    const u64 kg = J.koff + k0 + kc * 16;
    if (kg >= Kp) return; // the last 64-wide block of the padded K range may stick out (Kp is a multiple of 32)
    i8 *dst = J.planes + ((kg >> 5) * J.Op + o0 + oo) * RK + (kg & 31);
    const u64 plane_stride = J.Op * Kp;
    // residue: t = sum d_j (256^j mod m) (|t| <= 8 * 128 * 128) minus m times the nearest integer to t / m, all in
    // float, where these integers are exact.  For the odd moduli t / m is at least 1 / 510 away from a tie, forty times
    // the error of the float product, so |r| <= (m - 1) / 2 <= 127; for m = 256, t is the lowest digit itself and
    // r = t.  The dot product starts at 128, so that r + 128 is a byte that v_cvt_pk_u8_f32 converts and places in
    // one instruction; flipping the top bits of the four bytes takes the bias off again.
    // Written stage by stage over the lane's 16 words so that no instruction waits for the one before it; the dot
    // products are spelled out because the three-address form (accumulator from another register: no copy of the
    // bias per product) is not what the compiler picks for the builtin.
    typedef float v2f __attribute__((ext_vector_type(2)));
    i32 bias = 128;
    asm volatile("" : "+v"(bias));
#pragma unroll 1
    for (int i = 0; i < NP; i++) {
        const i32 clo = T.clo[i], chi = T.chi[i];
        const float w = T.w[i], mf = (float)T.m[i], wb = -128.f * w;
        const v2f w2 = {w, w}, wb2 = {wb, wb}, nm2 = {-mf, -mf};
        i32 t[16];
#pragma unroll
        for (u32 q = 0; q < 16; q++)
            asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(t[q]) : "v"((u32)y[q]), "s"(clo), "v"(bias));
#pragma unroll
        for (u32 q = 0; q < 16; q++)
            asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(t[q]) : "v"((u32)(y[q] >> 32)), "s"(chi), "v"(t[q]));
        // a dot product's result may be read by another kind of instruction three issue slots later at the
        // earliest; the compiler cannot see what the asm statements are, so the distance is made here
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 2");
        __builtin_amdgcn_sched_barrier(0);
        v2f tf[8], qf[8];
#pragma unroll
        for (u32 q = 0; q < 8; q++) {
            tf[q][0] = (float)t[2 * q];
            tf[q][1] = (float)t[2 * q + 1];
        }
#pragma unroll
        for (u32 q = 0; q < 8; q++) qf[q] = __builtin_elementwise_fma(tf[q], w2, wb2);
#pragma unroll
        for (u32 q = 0; q < 8; q++) {
            qf[q][0] = rintf(qf[q][0]);
            qf[q][1] = rintf(qf[q][1]);
        }
#pragma unroll
        for (u32 q = 0; q < 8; q++) tf[q] = __builtin_elementwise_fma(qf[q], nm2, tf[q]);
        u32 wd[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 e = 0; e < 4; e++)
#pragma unroll
            for (u32 d = 0; d < 4; d++) wd[d] = __builtin_amdgcn_cvt_pk_u8_f32(tf[2 * d + (e >> 1)][e & 1], e, wd[d]);
#pragma unroll
        for (u32 d = 0; d < 4; d++) wd[d] ^= 0x80808080u;
        st_stream((uint4 *)(dst + (u64)i * plane_stride), wd[0], wd[1], wd[2], wd[3]);
    }
}

// Y_i[m][n] = ((sum_k A_i[m][k] B_i[n][k]) mod m_i) * inv_i mod m_i as one byte, for modulus i = blockIdx.y and one
// 256 x 256 block tile.  Eight waves as 2(M) x 4(N); a wave owns 128 x 64 outputs = 4 x 2 MFMA tiles (8 accumulators).
// A stage is three K blocks (96 bytes of K) of the A and B tiles = 48 KiB, staged by LDS-DMA in 1-KiB pieces (six per
// wave and stage, issued singly between MFMAs) into three buffers: while stage t is multiplied, stage t+1 has landed
// or is landing and stage t+2 is being issued.  The XOR swizzle of the 16-byte halves is the positional kernel's.
__global__ __launch_bounds__(512) void rns_gemm_kernel(const i8 *__restrict__ Apl, const i8 *__restrict__ Bpl,
                                                       u8 *__restrict__ Y, u32 Mp, u32 Np, u32 nkb, u32 tiles_m, u32 tiles_n,
                                                       u32 GM, RnsTab T) {
    __shared__ __attribute__((aligned(1024))) i8 lds[3 * STAGE_BYTES];
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 wm = wid >> 2, wn = wid & 3, lr = lane & 31, lh = lane >> 5;
    const u32 mod_i = blockIdx.y;

    // XCD-aware order: workgroups b, b+8, .. share an L2; give each XCD a contiguous run of tiles, walked in
    // 4-row groups column by column
    const u32 nwg = tiles_m * tiles_n;
    const u32 bid = blockIdx.x;
    const u32 qq = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const u32 swz = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + bid / 8;
    const u32 per_group = GM * tiles_n;
    const u32 group = swz / per_group, in_group = swz % per_group;
    const u32 first_m = group * GM;
    const u32 gsz = (tiles_m - first_m < GM) ? tiles_m - first_m : GM;
    const u32 tm = first_m + in_group % gsz, tn = in_group / gsz;

    const i8 *Ai = Apl + (u64)mod_i * nkb * Mp * RK;
    const i8 *Bi = Bpl + (u64)mod_i * nkb * Np * RK;

    // staging plan: 48 pieces of 1 KiB per stage, 6 per wave: piece p -> K block p / 16, then A rows 32 (p % 16) ..
    // (p % 16 < 8) or B rows 32 (p % 16 - 8) ..
    const u32 lrow = lane >> 1;
    const u32 lsrc = lrow * RK + (((lane & 1) ^ ((lrow >> 3) & 1)) << 4);
    const i8 *gsrc[6];
    u32 gstep[6], ldst[6];
#pragma unroll
    for (u32 q = 0; q < 6; q++) {
        const u32 p = wid * 6 + q, kb = p >> 4, idx = p & 15;
        if (idx < 8) {
            gsrc[q] = Ai + ((u64)kb * Mp + (u64)tm * TM + idx * 32) * RK;
            gstep[q] = SKB * Mp * RK;
            ldst[q] = kb * KB_BYTES + idx * 1024;
        } else {
            gsrc[q] = Bi + ((u64)kb * Np + (u64)tn * TN + (idx - 8) * 32) * RK;
            gstep[q] = SKB * Np * RK;
            ldst[q] = kb * KB_BYTES + TM * RK + (idx - 8) * 1024;
        }
    }
    auto stage_piece = [&](u32 buf, u32 q) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[q] + lsrc),
                                         (__attribute__((address_space(3))) void *)(lds + buf * STAGE_BYTES + ldst[q]), 16,
                                         0, 0);
        gsrc[q] += gstep[q];
    };
    const u32 fsw = (lh ^ ((lr >> 3) & 1)) << 4;
    const u32 a_off = (wm * 128 + lr) * RK + fsw;
    const u32 b_off = TM * RK + (wn * 64 + lr) * RK + fsw;

    v16i acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;

    const u32 nst = nkb / SKB;
#pragma unroll
    for (u32 q = 0; q < 6; q++) stage_piece(0, q);
    if (nst > 1) {
#pragma unroll
        for (u32 q = 0; q < 6; q++) stage_piece(1, q);
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#define MPCX_PIN() __builtin_amdgcn_sched_barrier(0)
    auto rdA = [&](u32 buf, u32 s, u32 mi) { return *(const v4i *)(lds + buf * STAGE_BYTES + s * KB_BYTES + a_off + mi * 32 * RK); };
    auto rdB = [&](u32 buf, u32 s, u32 ni) { return *(const v4i *)(lds + buf * STAGE_BYTES + s * KB_BYTES + b_off + ni * 32 * RK); };
    // 8 MFMAs of one K block with up to three DMA pieces of stage t+2 between them (after the 2nd, 5th and 8th)
    auto block = [&](const v4i(&a)[4], const v4i(&b)[2], bool more, u32 fill, u32 q0, u32 nq) {
        u32 n = 0, q = 0;
#pragma unroll
        for (u32 mi = 0; mi < 4; mi++)
#pragma unroll
            for (u32 ni = 0; ni < 2; ni++) {
                acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                n++;
                if ((n == 2 || n == 5 || n == 8) && q < nq) {
                    MPCX_PIN();
                    if (more) stage_piece(fill, q0 + q);
                    MPCX_PIN();
                    q++;
                }
            }
    };
    // One barrier per stage, placed after the second K block: by then every wave holds all fragments of stage t and
    // its pieces of stage t+1 have landed, so the first fragments of stage t+1 are read while the third K block is
    // multiplied and the matrix pipe never waits for LDS right after a barrier.  Stage t+2 goes to the buffer of
    // stage t-1, free since the previous barrier.
    v4i a0[4], b0[2];
#pragma unroll
    for (u32 mi = 0; mi < 4; mi++) a0[mi] = rdA(0, 0, mi);
#pragma unroll
    for (u32 ni = 0; ni < 2; ni++) b0[ni] = rdB(0, 0, ni);
    u32 cur = 0;
    for (u32 t = 0; t < nst; t++) {
        const u32 nxt = cur == 2 ? 0 : cur + 1;
        const u32 fill = cur == 0 ? 2 : cur - 1;
        const bool more = t + 2 < nst;
        v4i a1[4], b1[2], a2[4], b2[2];
#pragma unroll
        for (u32 mi = 0; mi < 4; mi++) a1[mi] = rdA(cur, 1, mi);
#pragma unroll
        for (u32 ni = 0; ni < 2; ni++) b1[ni] = rdB(cur, 1, ni);
        block(a0, b0, more, fill, 0, 3);
#pragma unroll
        for (u32 mi = 0; mi < 4; mi++) a2[mi] = rdA(cur, 2, mi);
#pragma unroll
        for (u32 ni = 0; ni < 2; ni++) b2[ni] = rdB(cur, 2, ni);
        block(a1, b1, more, fill, 3, 3);
        // stage t+1's six pieces are older than the six just issued
        if (more)
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 1 < nst) {
#pragma unroll
            for (u32 mi = 0; mi < 4; mi++) a0[mi] = rdA(nxt, 0, mi);
#pragma unroll
            for (u32 ni = 0; ni < 2; ni++) b0[ni] = rdB(nxt, 0, ni);
        }
        block(a2, b2, more, fill, 6, 0);
        cur = nxt;
    }
#undef MPCX_PIN

    // epilogue: acc mod m, times inv, mod m -> one byte
    const u32 m = T.m[mod_i], inv = T.inv[mod_i], off = T.off_acc[mod_i];
    const u64 magic = T.magic40[mod_i];
    u8 *Yi = Y + (u64)mod_i * Mp * Np;
#pragma unroll
    for (u32 mi = 0; mi < 4; mi++)
#pragma unroll
        for (u32 ni = 0; ni < 2; ni++) {
            const u32 col = tn * TN + wn * 64 + ni * 32 + lr;
            const u32 row0 = tm * TM + wm * 128 + mi * 32 + 4 * lh;
#pragma unroll
            for (u32 r = 0; r < 16; r++) {
                const u32 row = row0 + (r & 3) + 8 * (r >> 2);
                u32 x = (u32)(acc[mi][ni][r] + (i32)off); // |acc| < 2^28 <= off
                x -= (u32)(((u64)x * magic) >> 40) * m;
                u32 yv = x * inv;
                yv -= (u32)(((u64)yv * magic) >> 40) * m;
                Yi[(u64)row * Np + col] = (u8)yv;
            }
        }
}

// The same product on a 384 x 256 block tile: eight waves as 2(M) x 4(N), a wave owns the even or the odd 32-row tiles
// of the block tile x 64 columns = 6 x 2 MFMA tiles (12 accumulators, 192 registers), so a K block stages 20 KiB for
// 96 MFMAs instead of 16 KiB for 64 -- a sixth fewer bytes through LDS-DMA per MFMA, which is what bounds the loop.
// One fragment set only: the A fragment of a row tile is read again for the next K block right after its two MFMAs
// were issued, the two B fragments alternate between two register pairs.  A stage is two K blocks = 40 KiB (five
// 1-KiB pieces per wave), three buffers; the barrier of a stage sits in the middle of its second K block, where the
// reads issued before it have had six MFMAs to return and six more cover the first reads from the next buffer.
//
// Rows are padded to 128 only: the last tile row of a product may hold 128 or 256 rows (NPW = 3 or 4 pieces per wave
// and stage instead of 5, NMI = 2 or 4 row tiles per wave instead of 6), so that no MFMA and no staged byte is spent
// on padding beyond 127 rows.
constexpr u32 TMW = 384, SKW = 2, MGRAN = 128;
constexpr u32 KBW_BYTES = (TMW + TN) * RK;    // 20 KiB
constexpr u32 STAGEW_BYTES = SKW * KBW_BYTES; // 40 KiB

#ifdef MPCX_DEV_PROBES
// Diagnostic build only (no stamp executes in the production library): per workgroup, the shader-clock ticks and the
// 100 MHz reference ticks that pass between the first staged piece and the last MFMA of its tile.  Their ratio is the
// clock the chip holds while the kernel runs.  Nothing in any kernel reads this array.
constexpr u32 RNS_CLOCK_SLOTS = 8192;
__device__ u64 rns_clock_probe[2 * RNS_CLOCK_SLOTS];
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // m0 on a clobber list: nothing else in the kernel uses it
template <u32 NPW>
__device__ __forceinline__ void rns_wide_tile(i8 *lds, const i8 *__restrict__ Ai, const i8 *__restrict__ Bi,
                                              u8 *__restrict__ Yi, u32 Mp, u32 Np, u32 nkb, u32 tm, u32 tn, u32 m,
                                              u32 inv, u32 off, u64 magic) {
    constexpr u32 NMI = 2 * NPW - 4;     // row tiles per wave: 6, 4, 2
    constexpr u32 VT = 2 * NMI;          // 32-row tiles of A in this block tile
    constexpr u32 PKB = VT + TN / 32;    // pieces per K block
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 wm = wid >> 2, wn = wid & 3, lr = lane & 31, lh = lane >> 5;

    // staging plan: NPW pieces of 1 KiB per wave and stage: piece p -> K block p / PKB, then A rows 32 (p % PKB) ..
    // (p % PKB < VT) or B rows 32 (p % PKB - VT) ..
    const u32 lrow = lane >> 1;
    const u32 lsrc = lrow * RK + (((lane & 1) ^ ((lrow >> 3) & 1)) << 4);
    const i8 *gsrc[NPW];
    u32 gstep[NPW], ldst[NPW];
#pragma unroll
    for (u32 q = 0; q < NPW; q++) {
        const u32 p = wid * NPW + q, kb = p / PKB, idx = p % PKB;
        if (idx < VT) {
            gsrc[q] = Ai + ((u64)kb * Mp + (u64)tm * TMW + idx * 32) * RK;
            gstep[q] = SKW * Mp * RK;
            ldst[q] = kb * KBW_BYTES + idx * 1024;
        } else {
            gsrc[q] = Bi + ((u64)kb * Np + (u64)tn * TN + (idx - VT) * 32) * RK;
            gstep[q] = SKW * Np * RK;
            ldst[q] = kb * KBW_BYTES + TMW * RK + (idx - VT) * 1024;
        }
    }
    // The DMA is issued from inline assembly (SGPR base + per-lane offset): with the builtin the compiler takes every
    // later LDS wait to zero outstanding reads instead of counting them, which stalls the in-place fragment reloads.
    const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) void *)lds;
    auto stage_piece = [&](u32 buf, u32 q) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lsrc), "s"(gsrc[q]), "s"(lds0 + buf * STAGEW_BYTES + ldst[q])
                     : "memory", "m0");
        gsrc[q] += gstep[q];
    };
    const u32 fsw = (lh ^ ((lr >> 3) & 1)) << 4;
    const u32 a_off = (wm * 32 + lr) * RK + fsw; // row tile 2 mi + wm
    const u32 b_off = TMW * RK + (wn * 64 + lr) * RK + fsw;

    v16i acc[NMI][2];
#pragma unroll
    for (u32 a = 0; a < NMI; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;

    const u32 nst = nkb / SKW;
#ifdef MPCX_DEV_PROBES
    const u64 stamp_clk0 = __builtin_amdgcn_s_memtime(), stamp_ref0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (u32 q = 0; q < NPW; q++) stage_piece(0, q);
    if (nst > 1) {
#pragma unroll
        for (u32 q = 0; q < NPW; q++) stage_piece(1, q);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#define MPCX_PIN() __builtin_amdgcn_sched_barrier(0)
    auto rdA = [&](u32 buf, u32 s, u32 mi) {
        return *(const v4i *)(lds + buf * STAGEW_BYTES + s * KBW_BYTES + a_off + mi * 64 * RK);
    };
    auto rdB = [&](u32 buf, u32 s, u32 ni) {
        return *(const v4i *)(lds + buf * STAGEW_BYTES + s * KBW_BYTES + b_off + ni * 32 * RK);
    };
    constexpr u32 H = NMI / 2; // row tiles of the second K block multiplied before the barrier
    v4i a[NMI], b[2], bn[2];
    // head: what follows a stage's barrier -- the first fragments of stage `cur` (landed: the barrier said so) are read
    // and, unless this is the very first stage, the second half of the previous stage's second K block is multiplied,
    // each A fragment replaced right after its two MFMAs.  tail: the same MFMAs with nothing left to read.
    auto head = [&](u32 cur, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        b[0] = rdB(cur, 0, 0);
        b[1] = rdB(cur, 0, 1);
#pragma unroll
        for (u32 mi = 0; mi < H; mi++) a[mi] = rdA(cur, 0, mi);
        MPCX_PIN();
#pragma unroll
        for (u32 mi = H; mi < NMI; mi++) {
            if (!FIRST) {
                acc[mi][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[0], acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[1], acc[mi][1], 0, 0, 0);
            }
            MPCX_PIN();
            a[mi] = rdA(cur, 0, mi);
            MPCX_PIN();
        }
    };
    auto tail = [&]() {
#pragma unroll
        for (u32 mi = H; mi < NMI; mi++) {
            acc[mi][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[0], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[1], acc[mi][1], 0, 0, 0);
        }
    };
    // body: the first K block of stage `cur`, the first half of its second K block and the stage's barrier; with MORE
    // the pieces of the stage after the next are issued in between (one after every second MFMA pair), into the buffer
    // of the previous stage, free since that stage's barrier
    auto body = [&](u32 cur, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        const u32 fill = cur == 0 ? 2 : cur - 1;
        u32 q = 0;
        bn[0] = rdB(cur, 1, 0);
        bn[1] = rdB(cur, 1, 1);
#pragma unroll
        for (u32 mi = 0; mi < NMI; mi++) {
            acc[mi][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[0], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[1], acc[mi][1], 0, 0, 0);
            MPCX_PIN();
            a[mi] = rdA(cur, 1, mi);
            if (MORE && (mi & 1) == 0 && q < NPW) stage_piece(fill, q++);
            MPCX_PIN();
        }
#pragma unroll
        for (u32 mi = 0; mi < H; mi++) {
            acc[mi][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[0], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], bn[1], acc[mi][1], 0, 0, 0);
            MPCX_PIN();
            if (MORE && (mi != 1 || NMI < 6) && q < NPW) stage_piece(fill, q++);
            MPCX_PIN();
        }
        if (MORE) {
#pragma unroll
            for (; q < NPW; q++) stage_piece(fill, q);
            // the next stage's pieces are older than the NPW just issued
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    u32 cur = 0, t = 0;
    head(0, std::true_type{});
    if (nst >= 3) {
        body(0, std::true_type{});
        cur = 1;
        for (t = 1; t + 2 < nst; t++) {
            head(cur, std::false_type{});
            body(cur, std::true_type{});
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
#pragma unroll 1
    for (; t < nst; t++) { // the last two stages: nothing left to issue
        if (t > 0) head(cur, std::false_type{});
        body(cur, std::false_type{});
        cur = cur == 2 ? 0 : cur + 1;
    }
    tail();
#undef MPCX_PIN
#ifdef MPCX_DEV_PROBES
    {
        const u64 stamp_clk1 = __builtin_amdgcn_s_memtime(), stamp_ref1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < RNS_CLOCK_SLOTS) {
            rns_clock_probe[2 * blockIdx.x] = stamp_clk1 - stamp_clk0;
            rns_clock_probe[2 * blockIdx.x + 1] = stamp_ref1 - stamp_ref0;
        }
    }
#endif

    // epilogue: acc mod m, times inv, mod m -> one byte
#pragma unroll
    for (u32 mi = 0; mi < NMI; mi++)
#pragma unroll
        for (u32 ni = 0; ni < 2; ni++) {
            const u32 col = tn * TN + wn * 64 + ni * 32 + lr;
            const u32 row0 = tm * TMW + (2 * mi + wm) * 32 + 4 * lh;
#pragma unroll
            for (u32 r = 0; r < 16; r++) {
                const u32 row = row0 + (r & 3) + 8 * (r >> 2);
                u32 x = (u32)(acc[mi][ni][r] + (i32)off); // |acc| < 2^28 <= off
                x -= (u32)(((u64)x * magic) >> 40) * m;
                u32 yv = x * inv;
                yv -= (u32)(((u64)yv * magic) >> 40) * m;
                Yi[(u64)row * Np + col] = (u8)yv;
            }
        }
}

#pragma clang diagnostic pop

// grid (tiles_m * tiles_n * NP); Mp is a multiple of 128, tiles_m = ceil(Mp / 384)
__global__ __launch_bounds__(512) void rns_gemm_wide_kernel(const i8 *__restrict__ Apl, const i8 *__restrict__ Bpl,
                                                            u8 *__restrict__ Y, u32 Mp, u32 Np, u32 nkb, u32 tiles_m,
                                                            u32 tiles_n, u32 GM, RnsTab T) {
    __shared__ __attribute__((aligned(1024))) i8 lds[3 * STAGEW_BYTES];
    // XCD-aware order over the whole launch: workgroups b, b+8, .. share an L2; every XCD takes one contiguous run of
    // the list (modulus, tile) -- about NP / 8 moduli -- so that the 32 tiles it works on at a time are one 4-row
    // group of one modulus' planes
    const u32 nwg = tiles_m * tiles_n, total = nwg * NP;
    const u32 bid = blockIdx.x;
    const u32 qq = total / 8, rem = total % 8, xcd = bid % 8;
    const u32 pos = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + bid / 8;
    const u32 mod_i = pos / nwg, swz = pos % nwg;
    const u32 per_group = GM * tiles_n;
    const u32 group = swz / per_group, in_group = swz % per_group;
    const u32 first_m = group * GM;
    const u32 gsz = (tiles_m - first_m < GM) ? tiles_m - first_m : GM;
    const u32 tm = first_m + in_group % gsz, tn = in_group / gsz;

    const i8 *Ai = Apl + (u64)mod_i * nkb * Mp * RK;
    const i8 *Bi = Bpl + (u64)mod_i * nkb * Np * RK;
    u8 *Yi = Y + (u64)mod_i * Mp * Np;
    const u32 m = T.m[mod_i], inv = T.inv[mod_i], off = T.off_acc[mod_i];
    const u64 magic = T.magic40[mod_i];
    const u32 rows = Mp - tm * TMW; // rows of this block tile that exist
    if (rows >= TMW)
        rns_wide_tile<5>(lds, Ai, Bi, Yi, Mp, Np, nkb, tm, tn, m, inv, off, magic);
    else if (rows == 2 * MGRAN)
        rns_wide_tile<4>(lds, Ai, Bi, Yi, Mp, Np, nkb, tm, tn, m, inv, off, magic);
    else
        rns_wide_tile<3>(lds, Ai, Bi, Yi, Mp, Np, nkb, tm, tn, m, inv, off, magic);
}

// Z[m][n] (+)= (sum_i y_i cM_i - round(sum_i y_i / m_i) Mmod) + add - sub, four columns per lane
__global__ __launch_bounds__(256) void rns_combine_kernel(const u8 *__restrict__ Y, u64 Mp, u64 Np, u64 M, u64 N,
                                                          const u64 *__restrict__ add, const u64 *__restrict__ sub,
                                                          u64 *__restrict__ Z, u64 ldz_row, u64 ldz_col, int accumulate,
                                                          RnsTab T) {
    const u64 nq = (N + 3) / 4, total = M * nq;
    const u64 plane = Mp * Np;
    for (u64 it = (u64)blockIdx.x * 256 + threadIdx.x; it < total; it += (u64)gridDim.x * 256) {
        const u64 row = it / nq, c0 = (it - row * nq) * 4;
        const u8 *p = Y + row * Np + c0; // Np is a multiple of 256: the four bytes are one aligned word
        u64 z[4] = {0, 0, 0, 0};
        float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const u32 w = *(const u32 *)(p + (u64)i * plane);
#pragma unroll
            for (u32 e = 0; e < 4; e++) {
                const u32 yv = (w >> (8 * e)) & 0xFFu;
                z[e] += (u64)yv * T.cM[i];
                f[e] = fmaf((float)yv, T.w[i], f[e]);
            }
        }
#pragma unroll
        for (u32 e = 0; e < 4; e++) {
            const u64 col = c0 + e;
            if (col >= N) break;
            u64 v = z[e] - (u64)(i64)rintf(f[e]) * T.Mmod;
            const u64 idx = row * N + col;
            if (add) v += add[idx];
            if (sub) v -= sub[idx];
            const u64 zi = row * ldz_row + col * ldz_col;
            if (accumulate) v += Z[zi];
            Z[zi] = v;
        }
    }
}

inline u64 round_up(u64 a, u64 b) { return ceil_div(a, b) * b; }
constexpr u64 KPAD = 64, KSTAGE = SKB * RK; // 96

u64 rns_part_chunk_len(int nparts, u64 l) {
    const u64 max_kc = 8192;
    u64 lc = round_up(l, KPAD);
    while ((u64)nparts * lc > max_kc && lc > KPAD) lc = round_up(lc / 2, KPAD);
    return lc;
}

} // namespace

// MPCX_GEMM_RNS: 0 = never, 2 = every shape, otherwise (default) by an estimate of both formulations' times from the
// rates measured on MI355X (ms; K = words along the contraction over all concatenated parts):
//   digit pairs: 128 x 64 output tiles, 256 at a time, 5.9e-5 per K and round of tiles (a single round that fills only
//                part of the chip runs at a higher clock: x (0.6 + 0.4 fill)), + 3.2e-9 per operand word for the digit
//                split -- 4096 x 8192 x 4096: 3.9, 2048^3: 0.275, 1024 x 2048 x 1024: 0.113;
//   residues   : 384 x 256 tiles (a last tile row of 128 / 256 rows costs 1/2 / 3/4) x 18 moduli over 256 CUs, + 0.35
//                rounds for the tail; 2.0e-5 per K + 0.006 (prologue, epilogue) per round; + 5.96e-9 per operand word
//                for the split + 7.5e-9 per output word for the CRT pass + 0.02 for the two extra launches --
//                4096 x 8192 x 4096: 2.62, 2048^3: 0.28, 1024 x 2048 x 1024: 0.092.
// The residue form wins from about 1024 x 2048 x 1024 when K is long and both output dimensions fill tiles; short K
// (4096 x 1024 x 4096: 0.55 against 0.52) and thin outputs (128 x 8192 x 8192: 0.68 against 0.55) stay with the digit
// pairs, and so does a tie (5 % margin).  Read on every call so that one process can exercise both paths.
bool ring_gemm_use_rns(int nparts, u64 r, u64 l, u64 c) {
    const char *e = getenv("MPCX_GEMM_RNS");
    const int mode = e ? atoi(e) : 1;
    if (mode == 0) return false;
    if (mode == 2) return true;
    if (r == 0 || l == 0 || c == 0) return false;
    const double K = (double)nparts * (double)l, words = (double)(r + c) * K;
    const double tiles_p = (double)ceil_div(r, 128) * (double)ceil_div(c, 64);
    const double fill = tiles_p < 256.0 ? 0.6 + 0.4 * tiles_p / 256.0 : 1.0;
    const double t_p = std::ceil(tiles_p / 256.0) * K * 5.9e-5 * fill + words * 3.2e-9;
    const u64 full = r / 384, rest = ceil_div(r % 384, 128); // whole tile rows, 128-row groups of the last one
    const double last = rest == 0 ? 0.0 : rest == 1 ? 0.5 : 0.75;
    const double cols = (double)ceil_div(c, 256);
    const double wgs = (double)(full + (rest ? 1 : 0)) * cols * 18.0;
    const double rounds_r = wgs <= 256.0 ? (full ? 1.0 : last) : ((double)full + last) * cols * 18.0 / 256.0 + 0.35;
    const double t_r = rounds_r * (K * 2.0e-5 + 0.006) + words * 5.96e-9 + (double)r * (double)c * 7.5e-9 + 0.02;
    return 1.05 * t_r < t_p;
}

// MPCX_RNS_TILE: 0 = 256 x 256 block tiles, otherwise (default) 384 x 256; read on every call
bool ring_gemm_rns_wide() {
    const char *e = getenv("MPCX_RNS_TILE");
    return e ? atoi(e) != 0 : true;
}

namespace {
struct RnsGeom {
    bool wide;
    u64 Mp, Np, lc, Kp;
};
RnsGeom rns_geom(int nparts, u64 r, u64 l, u64 c) {
    RnsGeom g;
    g.wide = ring_gemm_rns_wide();
    g.Mp = round_up(r, g.wide ? MGRAN : TM);
    g.Np = round_up(c, TN);
    g.lc = rns_part_chunk_len(nparts, l);
    g.Kp = round_up((u64)nparts * g.lc, g.wide ? SKW * RK : KSTAGE);
    return g;
}
} // namespace

u64 ring_gemm_rns_ws_bytes(int nparts, u64 r, u64 l, u64 c) {
    const RnsGeom g = rns_geom(nparts, r, l, c);
    const u64 a = (NP * g.Mp * g.Kp + 255) & ~255ull, b = (NP * g.Np * g.Kp + 255) & ~255ull,
              y = (NP * g.Mp * g.Np + 255) & ~255ull;
    return a + b + y;
}

// splits the second operands of one K chunk [kb, kb + klen) into the planes at Bpl
static void rns_split_b_chunk(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 c, u64 l, int layB, u64 kb, u64 klen,
                              const RnsGeom &geom, i8 *Bpl, const RnsTab &T) {
    const u64 soB = layB ? l : 1, skB = layB ? 1 : c;
    SplitJobs jobs;
    u32 n = 0;
    for (int p = 0; p < nparts; p++) {
        const u64 cover = p + 1 == nparts ? geom.Kp - (u64)p * geom.lc : geom.lc;
        jobs.j[n++] = SplitJob{(const u64 *)parts[p].B + kb * skB,
                               parts[p].Bplus ? (const u64 *)parts[p].Bplus + kb * skB : nullptr, Bpl, c, soB, skB, geom.Np,
                               (u64)p * geom.lc, cover};
    }
    const u32 gx = (u32)ceil_div(geom.Kp - (u64)(nparts - 1) * geom.lc, 64);
    if (skB == 1)
        hipLaunchKernelGGL(rns_split_kernel<true>, dim3(gx, (u32)(geom.Np / 64), n), dim3(256), 0, ctx->stream, jobs, klen,
                           geom.Kp, T);
    else
        hipLaunchKernelGGL(rns_split_kernel<false>, dim3(gx, (u32)(geom.Np / 64), n), dim3(256), 0, ctx->stream, jobs, klen,
                           geom.Kp, T);
}

int ring_gemm_rns_prepare_b(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 l, u64 c, int layB, RnsPreB *out) {
    MPCX_REQUIRE(l > 0 && c > 0 && l < (1u << 30) && c < (1u << 30), "ring gemm: bad operand shape");
    MPCX_REQUIRE(nparts >= 1 && nparts <= 8, "ring gemm: bad part count");
    const RnsTab &T = rns_tab();
    const RnsGeom geom = rns_geom(nparts, 1, l, c);
    MPCX_REQUIRE((u64)nparts * geom.lc <= 8192, "ring gemm: cannot chunk K");
    MPCX_REQUIRE(geom.Np / 64 <= 65535, "ring gemm: more than 4194240 columns");
    RnsPreB pre;
    pre.chunk_bytes = (NP * geom.Np * geom.Kp + 255) & ~255ull;
    pre.nchunks = ceil_div(l, geom.lc);
    pre.nparts = nparts, pre.l = l, pre.c = c, pre.wide = geom.wide;
    void *mem = nullptr;
    MPCX_TRY(mpcx_malloc(ctx, pre.nchunks * pre.chunk_bytes, &mem));
    pre.planes = (i8 *)mem;
    {
        ProfScope ps(ctx, PROF_SPLIT);
        u64 ci = 0;
        for (u64 kb = 0; kb < l; kb += geom.lc, ci++)
            rns_split_b_chunk(ctx, parts, nparts, c, l, layB, kb, (l - kb < geom.lc) ? l - kb : geom.lc, geom,
                              pre.planes + ci * pre.chunk_bytes, T);
    }
    if (hipGetLastError() != hipSuccess) {
        (void)mpcx_free(ctx, mem);
        MPCX_REQUIRE(false, "ring gemm: the split of a resident operand failed to launch");
    }
    *out = pre;
    return 0;
}
void ring_gemm_rns_release_b(mpcx_ctx *ctx, RnsPreB *pre) {
    if (pre && pre->planes) (void)mpcx_free(ctx, pre->planes);
    if (pre) *pre = RnsPreB();
}

int ring_gemm_parts_rns(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                        const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, const RnsPreB *pre) {
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0, "ring gemm: empty operand (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    MPCX_REQUIRE(r < (1u << 30) && c < (1u << 30) && l < (1u << 30), "ring gemm: dimension too large");
    MPCX_REQUIRE(nparts >= 1 && nparts <= 8, "ring gemm: bad part count");
    const RnsTab &T = rns_tab();
    const RnsGeom geom = rns_geom(nparts, r, l, c);
    const u64 Mp = geom.Mp, Np = geom.Np, lc = geom.lc, Kp = geom.Kp;
    MPCX_REQUIRE((u64)nparts * lc <= 8192, "ring gemm: cannot chunk K");
    // the split grid has one y block per 64 rows, and the kernels step through the planes with 32-bit byte offsets
    MPCX_REQUIRE(Mp / 64 <= 65535 && Np / 64 <= 65535, "ring gemm: more than 4194240 rows or columns");
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, ring_gemm_rns_ws_bytes(nparts, r, l, c), &ws));
    i8 *Apl = (i8 *)ws;
    i8 *Bpl = Apl + ((NP * Mp * Kp + 255) & ~255ull);
    u8 *Y = (u8 *)Bpl + ((NP * Np * Kp + 255) & ~255ull);
    const u32 tiles_m = (u32)(geom.wide ? ceil_div(Mp, TMW) : Mp / TM), tiles_n = (u32)(Np / TN);
    MPCX_REQUIRE((u64)tiles_m * tiles_n * NP < (1ull << 31), "ring gemm: too many output tiles for one launch");
    const u32 nkb = (u32)(Kp / RK);
    const u64 soA = layA ? 1 : l, skA = layA ? r : 1;
    const u64 soB = layB ? l : 1, skB = layB ? 1 : c;
    const u32 gm = gemm_group_rows("MPCX_RNS_GM", tiles_m); // tile rows per group (tuning)
    // planes of the second operand made earlier for exactly this geometry are read in place of a split of B
    const bool have_b = pre && pre->planes && pre->nparts == nparts && pre->l == l && pre->c == c && pre->wide == geom.wide &&
                        pre->chunk_bytes == ((NP * Np * Kp + 255) & ~255ull) && pre->nchunks == ceil_div(l, lc);
    if (have_b) pre->used++;
    i8 *const Bws = Bpl;
    int first = 1;
    u64 chunk = 0;
    for (u64 kb = 0; kb < l; kb += lc, chunk++) {
        const u64 klen = (l - kb < lc) ? l - kb : lc;
        Bpl = have_b ? pre->planes + chunk * pre->chunk_bytes : Bws;
        {
            ProfScope ps(ctx, PROF_SPLIT);
            SplitJobs fast, slow;
            u32 nfast = 0, nslow = 0;
            u64 fast_op = 0, slow_op = 0;
            for (int p = 0; p < nparts; p++) {
                // the last part also writes the zeros up to the stage-padded end of K
                const u64 cover = p + 1 == nparts ? Kp - (u64)p * lc : lc;
                const SplitJob ja{(const u64 *)parts[p].A + kb * skA, nullptr, Apl, r, soA, skA, Mp, (u64)p * lc, cover};
                const SplitJob jb{(const u64 *)parts[p].B + kb * skB,
                                  parts[p].Bplus ? (const u64 *)parts[p].Bplus + kb * skB : nullptr, Bpl, c, soB, skB, Np,
                                  (u64)p * lc, cover};
                const SplitJob both[2] = {ja, jb};
                for (int q = 0; q < (have_b ? 1 : 2); q++) {
                    const SplitJob &j = both[q];
                    if (j.sk == 1) {
                        fast.j[nfast++] = j;
                        if (j.Op > fast_op) fast_op = j.Op;
                    } else {
                        slow.j[nslow++] = j;
                        if (j.Op > slow_op) slow_op = j.Op;
                    }
                }
            }
            const u32 gx = (u32)ceil_div(Kp - (u64)(nparts - 1) * lc, 64);
            if (nfast)
                hipLaunchKernelGGL(rns_split_kernel<true>, dim3(gx, (u32)(fast_op / 64), nfast), dim3(256), 0, ctx->stream,
                                   fast, klen, Kp, T);
            if (nslow)
                hipLaunchKernelGGL(rns_split_kernel<false>, dim3(gx, (u32)(slow_op / 64), nslow), dim3(256), 0, ctx->stream,
                                   slow, klen, Kp, T);
        }
        {
            ProfScope ps(ctx, PROF_GEMM);
            if (geom.wide)
                hipLaunchKernelGGL(rns_gemm_wide_kernel, dim3(tiles_m * tiles_n * NP), dim3(512), 0, ctx->stream, Apl, Bpl, Y,
                                   (u32)Mp, (u32)Np, nkb, tiles_m, tiles_n, gm, T);
            else
                hipLaunchKernelGGL(rns_gemm_kernel, dim3(tiles_m * tiles_n, NP), dim3(512), 0, ctx->stream, Apl, Bpl, Y,
                                   (u32)Mp, (u32)Np, nkb, tiles_m, tiles_n, gm, T);
        }
        {
            ProfScope ps(ctx, PROF_SPLIT);
            const bool last = kb + lc >= l;
            const u64 *ad = last ? (const u64 *)add : nullptr, *sb = last ? (const u64 *)sub : nullptr;
            const u64 ldr = layZ ? (u64)1 : c, ldc = layZ ? r : (u64)1;
            const int acc = (accumulate || !first) ? 1 : 0;
            u64 blocks = ceil_div(r * ((c + 3) / 4), 256);
            if (blocks > MPCX_MAX_GRID) blocks = MPCX_MAX_GRID;
            hipLaunchKernelGGL(rns_combine_kernel, dim3((u32)blocks), dim3(256), 0, ctx->stream, Y, Mp, Np, r, c, ad, sb,
                               (u64 *)Z, ldr, ldc, acc, T);
        }
        first = 0;
    }
    MPCX_LAUNCH_CHECK();
    return 0;
}

#ifdef MPCX_DEV_PROBES
// host side of the clock probe: copies the first `slots` (shader ticks, reference ticks) pairs of the last launches
extern "C" int mpcx_dev_rns_clock_read(uint64_t *out, uint64_t slots) {
    if (!out || slots > RNS_CLOCK_SLOTS) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rns_clock_probe), 2 * slots * sizeof(u64)) == hipSuccess ? 0 : 1;
}
extern "C" int mpcx_dev_rns_clock_clear(void) {
    static u64 zeros[2 * RNS_CLOCK_SLOTS];
    return hipMemcpyToSymbol(HIP_SYMBOL(rns_clock_probe), zeros, sizeof zeros) == hipSuccess ? 0 : 1;
}
#endif
