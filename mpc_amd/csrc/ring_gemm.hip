// Exact Z_2^64 matrix product on the int8 matrix cores of gfx950.
//
// A 64-bit ring element is rewritten as eight SIGNED base-256 digits d_0..d_7 in [-128,127]
// (x = sum d_i 256^i mod 2^64; digit i of x is byte i of (x + 0x80..80) ^ 0x80..80).  Then
//     A*B mod 2^64 = sum_{s=0..7} 256^s * P_s ,   P_s = sum_{i+j=s} A_i * B_j   (36 int8 products)
// and every P_s fits an int32 accumulator as long as (s+1) * K * 2^14 < 2^31, i.e. K <= 16383 per
// accumulation chunk (larger K is split into chunks that are recombined in 64 bits).
//
// Pipeline:  split_limbs (u64 matrix -> 8 int8 limb planes, optionally two sources concatenated
// along K)  ->  limb_gemm (v_mfma_i32_32x32x32_i8; the eight P_s of an output element stay in the registers
// of one lane and are shift-added in 64 bits in the epilogue, which also adds the mask / accumulates).
//
// Limb planes are stored K-blocked: plane[i][kb][row][32 bytes], kb = k / 32, so that the operand
// tile of one K step (a run of rows x 32 bytes) is one contiguous, fully used piece of memory.
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "mpcx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr u32 RM = 128, RN = 64, RK = 32; // block tile: 128 x 64 outputs, 32 bytes of K per step
constexpr u32 KPAD = 64;                  // K extent of each concatenated part is padded to this
constexpr u32 STAGE_BYTES = 8 * (RM + RN) * RK; // 48 KiB: all 8 limbs of the A and B tiles of one step

// ---------------------------------------------------------------- split
// planes[i][kb][o][k%32] = digit_i( src(o,k) (+ src2(o,k)) ),  src(o,k) = src[o*so + k*sk], k offset by koff;
// everything outside [0,O) x [0,K) is written as 0.  One launch serves a table of sources that share K and Kp
// (the operands of all concatenated parts): grid (Kpart/64, max Op/64, jobs).
struct SplitJob {
    const u64 *src, *src2; // src2 (nullable) is added element-wise
    i8 *planes;
    u64 O, so, sk, Op, koff;
};
struct SplitJobs {
    SplitJob j[16];
};

// byte i of each of y[0..15] -> 16 consecutive bytes of limb plane i
__device__ inline void store_limbs(const u64 (&y)[16], i8 *dst, u64 plane_stride) {
#pragma unroll
    for (u32 i = 0; i < 8; i++) {
        u32 w[4];
#pragma unroll
        for (u32 d = 0; d < 4; d++) {
            w[d] = (u32)((y[4 * d] >> (8 * i)) & 0xFF) | (u32)((y[4 * d + 1] >> (8 * i)) & 0xFF) << 8 |
                   (u32)((y[4 * d + 2] >> (8 * i)) & 0xFF) << 16 | (u32)((y[4 * d + 3] >> (8 * i)) & 0xFF) << 24;
        }
        *(uint4 *)(dst + i * plane_stride) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// KFAST: k is the source's fast axis (sk == 1).  A lane's 16 consecutive k are then contiguous in memory, so the
// 64 x 64 tile is read with k along the lanes and turned through LDS.  Otherwise o is the fast axis: a lane reads
// its own 16 words (one per k row; a wave load covers two runs of 32 consecutive o) and nothing is exchanged.
template <bool KFAST>
__global__ __launch_bounds__(256) void split_limbs_kernel(SplitJobs jobs, u64 K, u64 Kp) {
    const SplitJob &J = jobs.j[blockIdx.z];
    const u32 tid = threadIdx.x;
    const u64 o0 = (u64)blockIdx.y * 64, k0 = (u64)blockIdx.x * 64;
    if (o0 >= J.Op) return;
    const u64 *__restrict__ src = J.src, *__restrict__ src2 = J.src2;
    const u64 O = J.O, so = J.so, sk = J.sk;
    // thread -> (row oo, 16 consecutive k): a wave takes 32 rows x both 16-byte halves of one 32-byte K block, so
    // each of its stores covers 1 KiB of contiguous plane memory
    const u32 oo = ((tid >> 1) & 31) | ((tid >> 7) << 5), kc = (tid & 1) | ((tid >> 6) & 1) << 1;
    u64 y[16];
    if constexpr (KFAST) {
        __shared__ u64 tile[64][65];
        // all 16 (or 32) loads of a lane are issued before the first is used: 32 KiB in flight per workgroup
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
#pragma unroll
    for (u32 q = 0; q < 16; q++) y[q] = (y[q] + 0x8080808080808080ull) ^ 0x8080808080808080ull;
    const u64 kg = J.koff + k0 + kc * 16;
    store_limbs(y, J.planes + ((kg >> 5) * J.Op + o0 + oo) * RK + (kg & 31), J.Op * Kp);
}

// ---------------------------------------------------------------- gemm
// Z[m][n] (+)= sum_s 256^s * sum_{i+j=s} sum_k Apl[i][m][k] * Bpl[j][n][k]  (+ add - sub) for one 128 x 64 block tile;
// add / sub are row-major M x N word arrays, Z is addressed as m * ldz_row + n * ldz_col.
//
// Eight waves as 4(M) x 2(N); every wave owns ONE 32x32 output tile and keeps all eight s-accumulators
// (8 x 16 registers).  Per K step the block stages the 8 limb tiles of A (8 x 128 x 32 B) and of B
// (8 x 64 x 32 B) once -- 48 KiB -- and every wave issues the 36 MFMAs i+j <= 7 on them: each staged
// byte feeds 192 multiply-adds (a plain one-accumulator 256x256 tiling feeds 128), and the accumulators
// never leave the registers: after the K loop each lane folds its eight int32 sums per element into one word.
//
// Staging is LDS-DMA (global_load_lds, 16 B per lane) into three 48 KiB buffers (steps t+1 and t+2 land while
// step t is multiplied).  A DMA wave-instruction writes 1 KiB linearly = 32 rows x 32 B, so the bank swizzle
// is applied to the per-lane SOURCE address and undone on the ds_read: 16-byte half h of row r is kept
// at half h ^ ((r >> 3) & 1), which spreads the 16 rows of a ds_read_b128 lane group over all 64 banks.
//
// Operand fragments: lane l supplies row (l & 31) and bytes 16*(l >> 5) .. +16 of the 32-byte step for
// both A and B, so every k meets its partner exactly once whatever the hardware's internal k order.
// Accumulator map (all gfx950 32x32 MFMAs): col = l & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
//
// ABL is 0 in every production build.  A build with -DMPCX_DEV_PROBES also instantiates the timing ablations
// 1 = no MFMA / fragment reads, 2 = no operand staging (their results are garbage).
template <int ABL>
__global__ __launch_bounds__(512) void limb_gemm_kernel(const i8 *__restrict__ Apl, const i8 *__restrict__ Bpl,
                                                        u64 *__restrict__ Z, u32 M, u32 N, u64 ldz_row, u64 ldz_col,
                                                        const u64 *__restrict__ add, const u64 *__restrict__ sub,
                                                        int accumulate, u32 Mp, u32 Np, u32 nkb, u32 tiles_m,
                                                        u32 tiles_n, u32 GM) {
    __shared__ __attribute__((aligned(1024))) i8 lds[3 * STAGE_BYTES]; // 144 KiB: three staging buffers
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 wm = wid >> 1, wn = wid & 1, lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile order.  Workgroups b, b+8, b+16, .. share an XCD and its L2: give each XCD one
    // contiguous run of tiles, and inside a run walk 4-row groups column by column so that the tiles in
    // flight together form a compact rectangle that re-uses the same A row panels and B column panels.
    const u32 nwg = tiles_m * tiles_n;
    const u32 bid = blockIdx.x;
    const u32 qq = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const u32 swz = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + bid / 8;
    const u32 per_group = GM * tiles_n;
    const u32 group = swz / per_group, in_group = swz % per_group;
    const u32 first_m = group * GM;
    const u32 gsz = (tiles_m - first_m < GM) ? tiles_m - first_m : GM;
    const u32 tm = first_m + in_group % gsz, tn = in_group / gsz;

    // ---- staging plan: 48 pieces of 1 KiB per step, 6 per wave.  Pieces 0..31: A limb p/4, rows 32*(p%4)..;
    // pieces 32..47: B limb (p-32)/2, rows 32*((p-32)%2)..
    const u32 lrow = lane >> 1;
    const u32 lsrc = lrow * RK + (((lane & 1) ^ ((lrow >> 3) & 1)) << 4); // swizzled source offset inside a piece
    const i8 *gsrc[6]; // wave-uniform piece bases (scalar registers); the lane offset lsrc is added at the load
    u32 gstep[6], ldst[6];
#pragma unroll
    for (u32 q = 0; q < 6; q++) {
        const u32 p = wid * 6 + q;
        if (p < 32) {
            const u32 limb = p >> 2, quarter = p & 3;
            gsrc[q] = Apl + ((u64)limb * nkb * Mp + (u64)tm * RM + quarter * 32) * RK;
            gstep[q] = Mp * RK;
            ldst[q] = limb * (RM * RK) + quarter * 1024;
        } else {
            const u32 limb = (p - 32) >> 1, half = (p - 32) & 1;
            gsrc[q] = Bpl + ((u64)limb * nkb * Np + (u64)tn * RN + half * 32) * RK;
            gstep[q] = Np * RK;
            ldst[q] = 8 * RM * RK + limb * (RN * RK) + half * 1024;
        }
    }
    // one DMA piece (1 KiB per wave instruction) of this wave's six
    auto stage_piece = [&](u32 buf, u32 q) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[q] + lsrc),
                                         (__attribute__((address_space(3))) void *)(lds + buf * STAGE_BYTES + ldst[q]), 16,
                                         0, 0);
        gsrc[q] += gstep[q];
    };

    // ---- fragment addresses (the swizzle term only depends on the lane)
    const u32 fsw = (lh ^ ((lr >> 3) & 1)) << 4;
    const u32 a_off = (wm * 32 + lr) * RK + fsw;
    const u32 b_off = 8 * RM * RK + (wn * 32 + lr) * RK + fsw;

    v16i acc[8];
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[s][r] = 0;

    // Software pipeline, three staging buffers, ONE barrier per K step placed in the middle of the step's MFMAs:
    //   first part : the 26 MFMAs with A limbs 0..3; the A fragments 1..7 of this step are read at its start;
    //   barrier    : every wave has all its fragments of step kb in registers and its DMA pieces of step kb+1
    //                have landed (counted vmcnt), so step kb+1 becomes visible to everyone;
    //   second part: the first fragments of step kb+1 are read while the last 10 MFMAs (A limbs 4..7) run from
    //                registers, so the matrix pipe never waits for LDS right after a barrier.
    // The six DMA pieces a wave issues for step kb+2 go to the buffer step kb-1 used (free since the previous
    // barrier).  A DMA piece costs its wave on the order of 100 issue cycles, so the pieces are spread singly
    // between MFMAs (four in the first part, two in the second) instead of being issued as one burst after the
    // barrier, where all eight waves would stall the matrix pipes together; each has more than half a step to land.
    auto rdA = [&](u32 buf, u32 i) { return *(const v4i *)(lds + buf * STAGE_BYTES + a_off + i * (RM * RK)); };
    auto rdB = [&](u32 buf, u32 j) { return *(const v4i *)(lds + buf * STAGE_BYTES + b_off + j * (RN * RK)); };
#define MPCX_PIN() __builtin_amdgcn_sched_barrier(0)
    v4i b[8], a0;
    if (ABL != 2) {
#pragma unroll
        for (u32 q = 0; q < 6; q++) stage_piece(0, q);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (nkb > 1) {
#pragma unroll
            for (u32 q = 0; q < 6; q++) stage_piece(1, q);
        }
    }
#pragma unroll
    for (u32 j = 0; j < 8; j++) b[j] = rdB(0, j);
    a0 = rdA(0, 0);
    u32 cur = 0;
    // one K step: multiplies with the fragments in (bu, au0), leaves the first fragments of the next step in
    // (bnx, anx0).  The two register sets swap roles every step (nkb is even), so no fragment is ever copied.
    // `more_tag`: whether a step kb+2 exists to be staged (true in the main loop, false in the two tail steps).
    auto step = [&](auto more_tag, const v4i(&bu)[8], const v4i &au0, v4i(&bnx)[8], v4i &anx0, bool last) {
        const u32 nxt = cur == 2 ? 0 : cur + 1;  // buffer of step kb+1
        const u32 fill = cur == 0 ? 2 : cur - 1; // buffer of step kb+2 (= of step kb-1)
        constexpr bool more = decltype(more_tag)::value && ABL != 2;
        v4i a[8];
        a[0] = au0;
#pragma unroll
        for (u32 i = 1; i < 8; i++) a[i] = rdA(cur, i);
        if (ABL != 1) {
            u32 m = 0;
#pragma unroll
            for (u32 i = 0; i < 4; i++)
#pragma unroll
                for (u32 j = 0; j + i < 8; j++) {
                    acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], bu[j], acc[i + j], 0, 0, 0);
                    m++;
                    if (m == 4 || m == 10 || m == 16 || m == 22) { // m is a compile-time value after unrolling
                        MPCX_PIN();
                        if (more) stage_piece(fill, (m - 4) / 6);
                        MPCX_PIN();
                    }
                }
        } else if (more) {
#pragma unroll
            for (u32 q = 0; q < 4; q++) stage_piece(fill, q);
        }
        // step kb+1's six pieces are older than the (up to four) just issued
        if (more)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!last) {
#pragma unroll
            for (u32 j = 0; j < 8; j++) bnx[j] = rdB(nxt, j);
            anx0 = rdA(nxt, 0);
        }
        if (ABL != 1) {
            u32 m = 0;
#pragma unroll
            for (u32 i = 4; i < 8; i++)
#pragma unroll
                for (u32 j = 0; j + i < 8; j++) {
                    acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], bu[j], acc[i + j], 0, 0, 0);
                    m++;
                    if (m == 3 || m == 7) {
                        MPCX_PIN();
                        if (more) stage_piece(fill, m == 3 ? 4 : 5);
                        MPCX_PIN();
                    }
                }
        } else if (more) {
            stage_piece(fill, 4);
            stage_piece(fill, 5);
        }
        cur = nxt;
    };
    v4i b2[8], a02;
#pragma unroll
    for (u32 j = 0; j < 8; j++) b2[j] = b[j];
    a02 = a0;
    // nkb is even (K is padded to a multiple of 64): pairs of steps, the last pair stages nothing
    for (u32 kb = 0; kb + 2 < nkb; kb += 2) {
        step(std::true_type{}, b, a0, b2, a02, false);
        step(std::true_type{}, b2, a02, b, a0, false);
    }
    step(std::false_type{}, b, a0, b2, a02, false);
    step(std::false_type{}, b2, a02, b, a0, true);
#undef MPCX_PIN
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- epilogue: one 64-bit word per element.  For a fixed register index the 64 lanes cover two rows x 32
    // consecutive columns, i.e. two 256-byte runs of a row-major Z.
    const u32 col = tn * RN + wn * 32 + lr;
    const u32 row0 = tm * RM + wm * 32 + 4 * lh;
    if (col < N) {
#pragma unroll
        for (u32 r = 0; r < 16; r++) {
            const u32 row = row0 + (r & 3) + 8 * (r >> 2);
            if (row >= M) continue;
            u64 z = 0;
#pragma unroll
            for (u32 s = 0; s < 8; s++) z += (u64)(i64)acc[s][r] << (8 * s);
            const u64 idx = (u64)row * N + col;
            if (add) z += add[idx];
            if (sub) z -= sub[idx];
            const u64 zi = row * ldz_row + col * ldz_col;
            if (accumulate) z += Z[zi];
            Z[zi] = z;
        }
    }
}


#ifdef MPCX_DEV_PROBES
// Development probe (MPCX_GEMM_ABLATE=3, results are garbage): the MFMA stream of limb_gemm_kernel re-expressed
// with v_mfma_i32_16x16x64_i8 (four per 32x32 tile and 64 bytes of K), fragments read from LDS, nothing staged.
// Only its run time and the clock the chip holds are of interest.
__global__ __launch_bounds__(512) void limb_gemm_shape_probe(i32 *__restrict__ Cpl, u32 Mp, u32 Np, u32 nkb) {
    __shared__ __attribute__((aligned(1024))) i8 lds[2 * STAGE_BYTES];
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 off = ((wid * 64 + lane) * 16) & (STAGE_BYTES - 1);
    v4i acc[8][4];
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[s][t] = v4i{0, 0, 0, 0};
    for (u32 kb = 0; kb < nkb; kb += 2) {
        v4i a[8][2], b[8][2];
#pragma unroll
        for (u32 i = 0; i < 8; i++)
#pragma unroll
            for (u32 h = 0; h < 2; h++) {
                a[i][h] = *(const v4i *)(lds + ((off + i * 4096 + h * 2048) & (2 * STAGE_BYTES - 16)));
                b[i][h] = *(const v4i *)(lds + ((off + 40960 + i * 2048 + h * 1024) & (2 * STAGE_BYTES - 16)));
            }
#pragma unroll
        for (u32 i = 0; i < 8; i++)
#pragma unroll
            for (u32 j = 0; j + i < 8; j++)
#pragma unroll
                for (u32 t = 0; t < 4; t++)
                    acc[i + j][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i][t >> 1], b[j][t & 1], acc[i + j][t], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    i32 *out = Cpl + ((u64)blockIdx.x * 512 + tid) * 4 % ((u64)Mp * Np);
    v4i sum = v4i{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) sum += acc[s][t];
    *(v4i *)out = sum;
}
#endif // MPCX_DEV_PROBES

// ---------------------------------------------------------------- relayout
__global__ __launch_bounds__(256) void relayout_kernel(const u64 *__restrict__ src, u64 *__restrict__ dst, u64 rows,
                                                       u64 cols, int lay_src, int lay_dst) {
    __shared__ u64 tile[32][33];
    const u32 tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const u64 r0 = (u64)blockIdx.y * 32, c0 = (u64)blockIdx.x * 32;
    for (u32 j = ty; j < 32; j += 8) {
        // read with the source's fast axis on tx
        const u64 r = lay_src ? r0 + tx : r0 + j, c = lay_src ? c0 + j : c0 + tx;
        if (r < rows && c < cols) tile[r - r0][c - c0] = src[lay_src ? c * rows + r : r * cols + c];
    }
    __syncthreads();
    for (u32 j = ty; j < 32; j += 8) {
        const u64 r = lay_dst ? r0 + tx : r0 + j, c = lay_dst ? c0 + j : c0 + tx;
        if (r < rows && c < cols) dst[lay_dst ? c * rows + r : r * cols + c] = tile[r - r0][c - c0];
    }
}

inline u64 round_up(u64 a, u64 b) { return ceil_div(a, b) * b; }

} // namespace

// Tile rows per group of the XCD-aware tile order (a tuning knob read from the environment, default 4), always
// within [1, tiles_m]: the kernels divide by it.
u32 gemm_group_rows(const char *env_name, u32 tiles_m) {
    const char *e = getenv(env_name);
    long v = e ? atol(e) : 4;
    if (v < 1) v = 1;
    if (v > (long)tiles_m) v = (long)tiles_m;
    return (u32)v;
}

// K chunking: per accumulation chunk (all parts concatenated) 8 * Kc * 2^14 must stay below 2^31
static u64 part_chunk_len(int nparts, u64 l) {
    const u64 max_kc = 8192;
    u64 lc = round_up(l, KPAD);
    while ((u64)nparts * lc > max_kc && lc > KPAD) lc = round_up(lc / 2, KPAD);
    return lc;
}

// limb planes of one accumulation chunk of the digit-pair kernels
static u64 positional_ws_bytes(int nparts, u64 r, u64 l, u64 c) {
    const u64 Mp = round_up(r, RM), Np = round_up(c, RM), Kp = (u64)nparts * part_chunk_len(nparts, l);
    const u64 bytesA = 8 * Mp * Kp, bytesB = 8 * Np * Kp;
    return ((bytesA + 255) & ~255ull) + ((bytesB + 255) & ~255ull);
}

u64 ring_gemm_ws_bytes(int nparts, u64 r, u64 l, u64 c) {
    if (r == 0 || l == 0 || c == 0) return 0;
    if (ring_gemm_use_rns(nparts, r, l, c)) return ring_gemm_rns_ws_bytes(nparts, r, l, c);
    return positional_ws_bytes(nparts, r, l, c);
}

static int ring_gemm_positional(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                                const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, void *ws);

// the product over an empty contraction (l = 0): every entry is the empty sum, Z (+)= add - sub
__global__ __launch_bounds__(256) void empty_sum_kernel(u64 r, u64 c, const u64 *__restrict__ add, const u64 *__restrict__ sub,
                                                        u64 *__restrict__ Z, u64 ldr, u64 ldc, int accumulate) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < r * c; i += (u64)gridDim.x * 256) {
        const u64 row = i / c, col = i - row * c, zi = row * ldr + col * ldc;
        u64 v = accumulate ? Z[zi] : 0;
        if (add) v += add[i];
        if (sub) v -= sub[i];
        Z[zi] = v;
    }
}

// Matrices without rows or columns are legal operands of the engines (Eigen multiplies them: an empty result, or all
// zeros over an empty contraction), so the protocol segments pass them through to here.
int ring_gemm_parts(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                    const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, const RnsPreB *pre) {
    if (r == 0 || c == 0) return 0;
    if (l == 0) {
        u64 blocks = ceil_div(r * c, 256);
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(empty_sum_kernel, dim3((u32)blocks), dim3(256), 0, ctx->stream, r, c, (const u64 *)add, (const u64 *)sub,
                           (u64 *)Z, layZ ? (u64)1 : c, layZ ? r : (u64)1, accumulate);
        MPCX_LAUNCH_CHECK();
        return 0;
    }
    if (ring_gemm_use_rns(nparts, r, l, c))
        return ring_gemm_parts_rns(ctx, parts, nparts, r, l, c, layA, layB, add, sub, Z, layZ, accumulate, pre);
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0, "ring gemm: empty operand (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, positional_ws_bytes(nparts, r, l, c), &ws));
    return ring_gemm_positional(ctx, parts, nparts, r, l, c, layA, layB, add, sub, Z, layZ, accumulate, ws);
}

// the digit-pair kernels on a caller-provided workspace of positional_ws_bytes(..) bytes, 256-byte aligned; every byte
// of the planes a chunk reads is written by that chunk's split launches, so the workspace needs no clearing
static int ring_gemm_positional(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                                const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, void *ws) {
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0, "ring gemm: empty operand (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    MPCX_REQUIRE(r < (1u << 30) && c < (1u << 30) && l < (1u << 30), "ring gemm: dimension too large");
    MPCX_REQUIRE(nparts >= 1 && nparts <= 8, "ring gemm: bad part count");
    // rows of both operands are padded to 128 (the split kernel works on 64-row tiles, the A tile is 128 rows)
    const u64 Mp = round_up(r, RM), Np = round_up(c, RM);
    const u64 lc = part_chunk_len(nparts, l); // per-part chunk length along l
    const u64 Kp = (u64)nparts * lc;
    MPCX_REQUIRE(Kp <= 8192, "ring gemm: cannot chunk K");
    MPCX_REQUIRE(Mp / 64 <= 65535 && Np / 64 <= 65535, "ring gemm: more than 4194240 rows or columns"); // split grid y
    const u64 bytesA = 8 * Mp * Kp;
    i8 *Apl = (i8 *)ws;
    i8 *Bpl = Apl + ((bytesA + 255) & ~255ull);
    const u32 tiles_m = (u32)(Mp / RM), tiles_n = (u32)(Np / RN);
    const u32 nkb = (u32)(Kp / RK);
    const u64 soA = layA ? 1 : l, skA = layA ? r : 1; // A(o=row, k)
    const u64 soB = layB ? l : 1, skB = layB ? 1 : c; // B(o=col, k)
#ifdef MPCX_DEV_PROBES
    static const int abl = getenv("MPCX_GEMM_ABLATE") ? atoi(getenv("MPCX_GEMM_ABLATE")) : 0;
#endif
    const u32 gm = gemm_group_rows("MPCX_GEMM_GM", tiles_m);
    int first = 1;
    for (u64 kb = 0; kb < l; kb += lc) {
        const u64 klen = (l - kb < lc) ? l - kb : lc;
        {
            ProfScope ps(ctx, PROF_SPLIT);
            SplitJobs fast, slow; // by access pattern: k contiguous in the source, or o contiguous
            u32 nfast = 0, nslow = 0;
            u64 fast_op = 0, slow_op = 0;
            for (int p = 0; p < nparts; p++) {
                const SplitJob ja{(const u64 *)parts[p].A + kb * skA, nullptr, Apl, r, soA, skA, Mp, (u64)p * lc};
                const SplitJob jb{(const u64 *)parts[p].B + kb * skB,
                                  parts[p].Bplus ? (const u64 *)parts[p].Bplus + kb * skB : nullptr, Bpl, c, soB, skB, Np,
                                  (u64)p * lc};
                for (const SplitJob &j : {ja, jb}) {
                    if (j.sk == 1) {
                        fast.j[nfast++] = j;
                        if (j.Op > fast_op) fast_op = j.Op;
                    } else {
                        slow.j[nslow++] = j;
                        if (j.Op > slow_op) slow_op = j.Op;
                    }
                }
            }
            if (nfast)
                hipLaunchKernelGGL(split_limbs_kernel<true>, dim3((u32)(lc / 64), (u32)(fast_op / 64), nfast), dim3(256), 0,
                                   ctx->stream, fast, klen, Kp);
            if (nslow)
                hipLaunchKernelGGL(split_limbs_kernel<false>, dim3((u32)(lc / 64), (u32)(slow_op / 64), nslow), dim3(256), 0,
                                   ctx->stream, slow, klen, Kp);
        }
        {
            ProfScope ps(ctx, PROF_GEMM);
            const dim3 grid(tiles_m * tiles_n), block(512);
            const bool last = kb + lc >= l;
            const u64 *ad = last ? (const u64 *)add : nullptr, *sb = last ? (const u64 *)sub : nullptr;
            const u64 ldr = layZ ? (u64)1 : c, ldc = layZ ? r : (u64)1;
            const int acc = (accumulate || !first) ? 1 : 0;
#define MPCX_GEMM_ARGS Apl, Bpl, (u64 *)Z, (u32)r, (u32)c, ldr, ldc, ad, sb, acc, (u32)Mp, (u32)Np, nkb, tiles_m, tiles_n, gm
#ifdef MPCX_DEV_PROBES
            if (abl == 3) // timing probe only: scribbles over Z
                hipLaunchKernelGGL(limb_gemm_shape_probe, grid, block, 0, ctx->stream, (i32 *)Z, (u32)r, (u32)c, nkb);
            else if (abl == 1)
                hipLaunchKernelGGL(limb_gemm_kernel<1>, grid, block, 0, ctx->stream, MPCX_GEMM_ARGS);
            else if (abl == 2)
                hipLaunchKernelGGL(limb_gemm_kernel<2>, grid, block, 0, ctx->stream, MPCX_GEMM_ARGS);
            else
#endif
                hipLaunchKernelGGL(limb_gemm_kernel<0>, grid, block, 0, ctx->stream, MPCX_GEMM_ARGS);
#undef MPCX_GEMM_ARGS
        }
        first = 0;
    }
    MPCX_LAUNCH_CHECK();
    return 0;
}

// ---- the entry point that allocates nothing: limb planes in a file-scope device array.  Every GPU holds its own
// copy of the array at its own address; the address is looked up once per device, with that device current, and
// only that pointer (never the array's name) is handed to kernels.
constexpr u64 STATIC_SCRATCH_BYTES = 16ull << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
static void *g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mu;

int ring_gemm_static_scratch_init(int device) {
    mpcx_ctx *ctx = nullptr; // errors of a context that does not exist yet go to the init error slot
    MPCX_REQUIRE(device >= 0 && device < MAX_DEVICES, "mpcx_init: device %d beyond the %d the static scratch table holds",
                 device, MAX_DEVICES);
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (g_scratch_ptr[device]) return 0;
    MPCX_HIP(hipSetDevice(device));
    void *ptr = nullptr;
    MPCX_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    MPCX_REQUIRE(ptr && ((uintptr_t)ptr & 255) == 0, "mpcx_init: static scratch of device %d at %p is not 256-byte aligned",
                 device, ptr);
    g_scratch_ptr[device] = ptr;
    return 0;
}

extern "C" {

int mpcx_ring_gemm_u64_noalloc(mpcx_ctx *ctx, const int64_t *A, const int64_t *B, int64_t *C, uint64_t r, uint64_t l,
                               uint64_t c, int layoutA, int layoutB, int layoutC, int accumulate) {
    MPCX_REQUIRE(ctx, "mpcx_ring_gemm_u64_noalloc: the context is NULL");
    MPCX_REQUIRE(A && B && C, "mpcx_ring_gemm_u64_noalloc: NULL operand");
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0 && r < (1u << 30) && c < (1u << 30) && l < (1u << 30),
                 "mpcx_ring_gemm_u64_noalloc: bad shape (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    const u64 need = positional_ws_bytes(1, r, l, c);
    MPCX_REQUIRE(need <= STATIC_SCRATCH_BYTES,
                 "mpcx_ring_gemm_u64_noalloc: a %llu x %llu . %llu x %llu product needs %llu bytes of limb planes, the "
                 "static scratch holds %llu (use mpcx_ring_gemm_u64)",
                 (unsigned long long)r, (unsigned long long)l, (unsigned long long)l, (unsigned long long)c,
                 (unsigned long long)need, (unsigned long long)STATIC_SCRATCH_BYTES);
    void *ws = g_scratch_ptr[ctx->device]; // looked up by mpcx_init of this device
    MPCX_REQUIRE(ws, "mpcx_ring_gemm_u64_noalloc: the static scratch of device %d was not set up", ctx->device);
    const bool prof = ctx->prof; // the profiling scopes create HIP events, which may allocate
    ctx->prof = false;
    GemmPart p{A, B, nullptr};
    const int rc = ring_gemm_positional(ctx, &p, 1, r, l, c, layoutA, layoutB, nullptr, nullptr, C, layoutC, accumulate, ws);
    ctx->prof = prof;
    return rc;
}

uint64_t mpcx_ring_gemm_noalloc_limit(void) { return STATIC_SCRATCH_BYTES; }
uint64_t mpcx_ring_gemm_noalloc_bytes(uint64_t r, uint64_t l, uint64_t c) { return positional_ws_bytes(1, r, l, c); }

int mpcx_ring_gemm_u64(mpcx_ctx *ctx, const int64_t *A, const int64_t *B, int64_t *C, uint64_t r, uint64_t l,
                       uint64_t c, int layoutA, int layoutB, int layoutC, int accumulate) {
    MPCX_REQUIRE(A && B && C, "mpcx_ring_gemm_u64: NULL operand");
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0, "ring gemm: empty operand (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    GemmPart p{A, B, nullptr};
    return ring_gemm_parts(ctx, &p, 1, r, l, c, layoutA, layoutB, nullptr, nullptr, C, layoutC, accumulate);
}

int mpcx_ring_gemm_products(uint64_t r, uint64_t l, uint64_t c) { return ring_gemm_use_rns(1, r, l, c) ? 18 : 36; }
const char *mpcx_ring_gemm_kernel_name(uint64_t r, uint64_t l, uint64_t c) {
    if (!ring_gemm_use_rns(1, r, l, c)) return "limb_gemm_kernel";
    return ring_gemm_rns_wide() ? "rns_gemm_wide_kernel" : "rns_gemm_kernel";
}

int mpcx_relayout(mpcx_ctx *ctx, const int64_t *src, int64_t *dst, uint64_t rows, uint64_t cols, int lay_src,
                  int lay_dst) {
    MPCX_REQUIRE(src && dst && src != dst, "mpcx_relayout: bad pointers (in-place is not supported)");
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(relayout_kernel, dim3((u32)ceil_div(cols, 32), (u32)ceil_div(rows, 32)), dim3(256), 0,
                       ctx->stream, (const u64 *)src, (u64 *)dst, rows, cols, lay_src, lay_dst);
    MPCX_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
