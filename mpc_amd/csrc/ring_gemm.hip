// Exact Z_2^64 matrix product on the int8 matrix cores of gfx950.
//
// A 64-bit ring element is rewritten as eight SIGNED base-256 digits d_0..d_7 in [-128,127]
// (x = sum d_i 256^i mod 2^64; digit i of x is byte i of (x + 0x80..80) ^ 0x80..80).  Then
//     A*B mod 2^64 = sum_{s=0..7} 256^s * P_s ,   P_s = sum_{i+j=s} A_i * B_j   (36 int8 products)
// and every P_s fits an int32 accumulator as long as (s+1) * K * 2^14 < 2^31, i.e. K <= 16383 per
// accumulation chunk (larger K is split into chunks that are recombined in 64 bits).
//
// Pipeline:  split_limbs (u64 matrix -> 8 int8 planes, K contiguous, optionally two sources
// concatenated along K)  ->  limb_gemm (MFMA v_mfma_i32_32x32x32_i8, LDS-staged tiles, one int32
// plane per s)  ->  combine (shift-add the planes in 64 bits, fuse the PRG mask / accumulation).
#include "mpcx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr u32 BM = 128, BN = 128, BK = 128; // block tile (BK in int8 = bytes)
constexpr u32 LDS_PITCH = BK + 16;          // 144 B: consecutive rows land on distinct 16-B bank slots

// ---------------------------------------------------------------- split
// planes[i][o][koff + k] = digit_i( src(o,k) (+ src2(o,k)) ),  src(o,k) = src[o*so + k*sk]
// grid: (ceil(Kpart/64), Op/64); everything outside [0,O) x [0,K) is written as 0.
__global__ __launch_bounds__(256) void split_limbs_kernel(const u64 *__restrict__ src, const u64 *__restrict__ src2,
                                                          u64 O, u64 K, u64 so, u64 sk, i8 *__restrict__ planes,
                                                          u64 Op, u64 Kp, u64 koff) {
    __shared__ u64 tile[64][65];
    const u32 tid = threadIdx.x;
    const u64 o0 = (u64)blockIdx.y * 64, k0 = (u64)blockIdx.x * 64;
    const bool k_fast = (sk == 1);
#pragma unroll 4
    for (u32 j = 0; j < 16; j++) {
        const u32 a = tid & 63, b = (tid >> 6) + 4 * j;
        const u32 ko = k_fast ? a : b, oo = k_fast ? b : a;
        const u64 o = o0 + oo, k = k0 + ko;
        u64 v = 0;
        if (o < O && k < K) {
            v = src[o * so + k * sk];
            if (src2) v += src2[o * so + k * sk];
        }
        tile[oo][ko] = v;
    }
    __syncthreads();
    const u32 oo = tid >> 2, kc = tid & 3;
    u64 y[16];
#pragma unroll
    for (u32 q = 0; q < 16; q++)
        y[q] = (tile[oo][kc * 16 + q] + 0x8080808080808080ull) ^ 0x8080808080808080ull;
    const u64 plane_stride = Op * Kp;
    i8 *dst = planes + (o0 + oo) * Kp + koff + k0 + kc * 16;
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

// ---------------------------------------------------------------- gemm
// Cpl[s][m][n] = sum_{i<=s} sum_k Apl[i][m][k] * Bpl[s-i][n][k]   (int32), for the block's 128x128 tile.
// 4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of 32x32.  Operand fragments: lane l supplies row
// (l & 31) and the 16 bytes k = 16*(l >> 5) .. +16 of a 32-deep K step for both A and B, so each k
// is multiplied exactly once whatever the hardware's internal k order.  Accumulator map (all gfx950
// 32x32 MFMAs): col = l & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
__global__ __launch_bounds__(256) void limb_gemm_kernel(const i8 *__restrict__ Apl, const i8 *__restrict__ Bpl,
                                                        i32 *__restrict__ Cpl, u32 Mp, u32 Np, u32 Kp, u32 tiles_m,
                                                        u32 tiles_n) {
    __shared__ __attribute__((aligned(16))) i8 lds[(BM + BN) * LDS_PITCH];
    i8 *As = lds, *Bs = lds + BM * LDS_PITCH;
    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 wm = wid >> 1, wn = wid & 1, lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: workgroups b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous run of tiles so that neighbours reuse the same A row-panel / B column-panels.
    const u32 nwg = tiles_m * tiles_n;
    const u32 bid = blockIdx.x;
    const u32 q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const u32 swz = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const u32 tm = swz / tiles_n, tn = swz % tiles_n;

    const u64 strideA = (u64)Mp * Kp, strideB = (u64)Np * Kp, strideC = (u64)Mp * Np;
    const i8 *Ablk = Apl + (u64)tm * BM * Kp;
    const i8 *Bblk = Bpl + (u64)tn * BN * Kp;

    for (u32 s = 0; s < 8; s++) {
        v16i acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0;
        for (u32 i = 0; i <= s; i++) {
            const i8 *Ap = Ablk + i * strideA;
            const i8 *Bp = Bblk + (s - i) * strideB;
            for (u32 k0 = 0; k0 < Kp; k0 += BK) {
                uint4 ra[4], rb[4];
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    const u32 c = tid + 256 * p, row = c >> 3, seg = c & 7;
                    ra[p] = *(const uint4 *)(Ap + (u64)row * Kp + k0 + seg * 16);
                    rb[p] = *(const uint4 *)(Bp + (u64)row * Kp + k0 + seg * 16);
                }
                __syncthreads(); // previous tile fully consumed
#pragma unroll
                for (u32 p = 0; p < 4; p++) {
                    const u32 c = tid + 256 * p, row = c >> 3, seg = c & 7;
                    *(uint4 *)(As + row * LDS_PITCH + seg * 16) = ra[p];
                    *(uint4 *)(Bs + row * LDS_PITCH + seg * 16) = rb[p];
                }
                __syncthreads();
#pragma unroll
                for (u32 kk = 0; kk < BK / 32; kk++) {
                    v4i a[2], b[2];
#pragma unroll
                    for (u32 t = 0; t < 2; t++) {
                        a[t] = *(const v4i *)(As + (wm * 64 + t * 32 + lr) * LDS_PITCH + kk * 32 + lh * 16);
                        b[t] = *(const v4i *)(Bs + (wn * 64 + t * 32 + lr) * LDS_PITCH + kk * 32 + lh * 16);
                    }
#pragma unroll
                    for (u32 mi = 0; mi < 2; mi++)
#pragma unroll
                        for (u32 ni = 0; ni < 2; ni++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
        i32 *Cs = Cpl + s * strideC;
#pragma unroll
        for (u32 mi = 0; mi < 2; mi++)
#pragma unroll
            for (u32 ni = 0; ni < 2; ni++)
#pragma unroll
                for (u32 r = 0; r < 16; r++) {
                    const u32 row = tm * BM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const u32 col = tn * BN + wn * 64 + ni * 32 + lr;
                    Cs[(u64)row * Np + col] = acc[mi][ni][r];
                }
    }
}

// ---------------------------------------------------------------- gemm, large tiles
// Same contraction on a 256x256 block tile: 8 waves as 2(M) x 4(N), each 128x64 = 4x2 MFMA tiles.
// K is walked as ONE flat sequence of 128-byte steps over the 36 (A plane i, B plane s-i) pairs; the
// accumulators are flushed to plane s whenever s changes.  Operand tiles go global -> LDS with
// global_load_lds (16 B per lane, no VGPR staging) into two 64 KiB buffers: the loads of step t+1 are
// in flight while step t is multiplied.  An LDS-DMA wave-instruction writes 1 KiB linearly (8 rows of
// 128 B), so the bank swizzle is applied to the per-lane SOURCE address and undone on the ds_read:
// 16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7), which puts the 16 rows a
// ds_read_b128 lane group touches on 16 different 16-byte bank slots.
constexpr u32 TM = 256, TN = 256, TK = 128;
__global__ __launch_bounds__(512) void limb_gemm256_kernel(const i8 *__restrict__ Apl, const i8 *__restrict__ Bpl,
                                                           i32 *__restrict__ Cpl, u32 Mp, u32 Np, u32 Kp, u32 tiles_m,
                                                           u32 tiles_n) {
    __shared__ __attribute__((aligned(1024))) i8 lds[2 * (TM + TN) * TK]; // 128 KiB
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 wm = wid >> 2, wn = wid & 3, lr = lane & 31, lh = lane >> 5;

    const u32 nwg = tiles_m * tiles_n;
    const u32 bid = blockIdx.x;
    const u32 qq = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const u32 swz = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + bid / 8;
    const u32 tm = swz / tiles_n, tn = swz % tiles_n;

    const u64 strideA = (u64)Mp * Kp, strideB = (u64)Np * Kp, strideC = (u64)Mp * Np;
    const i8 *Ablk = Apl + (u64)tm * TM * Kp;
    const i8 *Bblk = Bpl + (u64)tn * TN * Kp;

    // staging: wave w issues 4 A pieces and 4 B pieces per step; piece p = 4w+q covers rows 8p..8p+7
    u32 goff[4];
#pragma unroll
    for (u32 q = 0; q < 4; q++) {
        const u32 row = (wid * 4 + q) * 8 + (lane >> 3);
        const u32 ch = (lane & 7) ^ ((row >> 1) & 7);
        goff[q] = row * Kp + ch * 16;
    }
    // fragment reads: row (within the wave's 128 / 64 rows) lr + 32*tile; the swizzle term only depends on lr
    const u32 rsw = (lr >> 1) & 7;
    const u32 a_row_off = (wm * 128 + lr) * TK, b_row_off = (wn * 64 + lr) * TK;

    const u32 nk = Kp / TK;
    const u32 T = 36 * nk;

    auto stage = [&](u32 buf, u32 pi, u32 pj, u32 kt) {
        const i8 *Ap = Ablk + pi * strideA + kt * TK;
        const i8 *Bp = Bblk + pj * strideB + kt * TK;
        i8 *la = lds + buf * ((TM + TN) * TK) + wid * 4096;
        i8 *lb = la + TM * TK;
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ap + goff[q]),
                                             (__attribute__((address_space(3))) void *)(la + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Bp + goff[q]),
                                             (__attribute__((address_space(3))) void *)(lb + q * 1024), 16, 0, 0);
        }
    };

    v16i acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;

    u32 ps = 0, pi = 0, pkt = 0; // tile being prefetched
    u32 cs = 0, ci = 0, ckt = 0; // tile being multiplied
    stage(0, 0, 0, 0);
    __syncthreads();
    u32 cur = 0;
    for (u32 t = 0; t < T; t++) {
        if (t + 1 < T) {
            if (++pkt == nk) {
                pkt = 0;
                if (pi == ps) {
                    ps++;
                    pi = 0;
                } else {
                    pi++;
                }
            }
            stage(cur ^ 1, pi, ps - pi, pkt);
        }
        const i8 *la = lds + cur * ((TM + TN) * TK);
        const i8 *lb = la + TM * TK;
#pragma unroll
        for (u32 kk = 0; kk < TK / 32; kk++) {
            const u32 coff = ((2 * kk + lh) ^ rsw) << 4;
            v4i a[4], b[2];
#pragma unroll
            for (u32 mi = 0; mi < 4; mi++) a[mi] = *(const v4i *)(la + a_row_off + mi * 32 * TK + coff);
#pragma unroll
            for (u32 ni = 0; ni < 2; ni++) b[ni] = *(const v4i *)(lb + b_row_off + ni * 32 * TK + coff);
#pragma unroll
            for (u32 mi = 0; mi < 4; mi++)
#pragma unroll
                for (u32 ni = 0; ni < 2; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        // last step of this s: flush the accumulators to plane s
        const bool last_k = (ckt + 1 == nk);
        if (last_k && ci == cs) {
            i32 *Cs = Cpl + cs * strideC;
#pragma unroll
            for (u32 mi = 0; mi < 4; mi++)
#pragma unroll
                for (u32 ni = 0; ni < 2; ni++)
#pragma unroll
                    for (u32 r = 0; r < 16; r++) {
                        const u32 row = tm * TM + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const u32 col = tn * TN + wn * 64 + ni * 32 + lr;
                        Cs[(u64)row * Np + col] = acc[mi][ni][r];
                        acc[mi][ni][r] = 0;
                    }
        }
        if (last_k) {
            ckt = 0;
            if (ci == cs) {
                cs++;
                ci = 0;
            } else {
                ci++;
            }
        } else {
            ckt++;
        }
        __syncthreads(); // drains this step's LDS-DMA (vmcnt) and the fragment reads before the buffers swap
        cur ^= 1;
    }
}

// ---------------------------------------------------------------- combine
// Z[m][n] (+)= sum_s (int64)Cpl[s][m][n] << 8s  (+ add[idx] - sub[idx]), idx = m*N + n (row-major)
__global__ __launch_bounds__(256) void combine_kernel(const i32 *__restrict__ Cpl, u64 Mp, u64 Np, u64 M, u64 N,
                                                      const u64 *__restrict__ add, const u64 *__restrict__ sub,
                                                      u64 *__restrict__ Z, u64 ldz_row, u64 ldz_col, int accumulate) {
    const u64 total = M * N;
    const u64 strideC = Mp * Np;
    for (u64 idx = (u64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (u64)gridDim.x * 256) {
        const u64 m = idx / N, n = idx - m * N;
        const i32 *p = Cpl + m * Np + n;
        u64 z = 0;
#pragma unroll
        for (u32 s = 0; s < 8; s++) z += (u64)(i64)p[s * strideC] << (8 * s);
        if (add) z += add[idx];
        if (sub) z -= sub[idx];
        const u64 zi = m * ldz_row + n * ldz_col;
        if (accumulate) z += Z[zi];
        Z[zi] = z;
    }
}

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

// K chunking: per accumulation chunk (all parts concatenated) 8 * Kc * 2^14 must stay below 2^31
static u64 part_chunk_len(int nparts, u64 l) {
    const u64 max_kc = 8192;
    u64 lc = round_up(l, BK);
    while ((u64)nparts * lc > max_kc && lc > BK) lc = round_up(lc / 2, BK);
    return lc;
}
// block tile edge: the 256-square LDS-DMA kernel for large outputs, the 128-square one otherwise
static u64 tile_edge(u64 r, u64 c) { return (r >= 512 && c >= 512) ? TM : BM; }

u64 ring_gemm_ws_bytes(int nparts, u64 r, u64 l, u64 c) {
    const u64 te = tile_edge(r, c);
    const u64 Mp = round_up(r, te), Np = round_up(c, te), Kp = (u64)nparts * part_chunk_len(nparts, l);
    const u64 bytesA = 8 * Mp * Kp, bytesB = 8 * Np * Kp, bytesC = 8 * Mp * Np * 4;
    return ((bytesA + 255) & ~255ull) + ((bytesB + 255) & ~255ull) + ((bytesC + 255) & ~255ull);
}

int ring_gemm_parts(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                    const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate) {
    MPCX_REQUIRE(r > 0 && l > 0 && c > 0, "ring gemm: empty operand (r=%llu l=%llu c=%llu)", (unsigned long long)r,
                 (unsigned long long)l, (unsigned long long)c);
    MPCX_REQUIRE(r < (1u << 30) && c < (1u << 30) && l < (1u << 30), "ring gemm: dimension too large");
    MPCX_REQUIRE(nparts >= 1 && nparts <= 8, "ring gemm: bad part count");
    const u64 te = tile_edge(r, c);
    const u64 Mp = round_up(r, te), Np = round_up(c, te);
    const u64 lc = part_chunk_len(nparts, l); // per-part chunk length along l
    const u64 Kp = (u64)nparts * lc;
    MPCX_REQUIRE(Kp <= 8192, "ring gemm: cannot chunk K");
    const u64 bytesA = 8 * Mp * Kp, bytesB = 8 * Np * Kp;
    void *ws;
    MPCX_TRY(mpcx_ws(ctx, ring_gemm_ws_bytes(nparts, r, l, c), &ws));
    i8 *Apl = (i8 *)ws;
    i8 *Bpl = Apl + ((bytesA + 255) & ~255ull);
    i32 *Cpl = (i32 *)(Bpl + ((bytesB + 255) & ~255ull));
    const u32 tiles_m = (u32)(Mp / te), tiles_n = (u32)(Np / te);
    const u64 soA = layA ? 1 : l, skA = layA ? r : 1; // A(o=row, k)
    const u64 soB = layB ? l : 1, skB = layB ? 1 : c; // B(o=col, k)
    int first = 1;
    for (u64 kb = 0; kb < l; kb += lc) {
        const u64 klen = (l - kb < lc) ? l - kb : lc;
        for (int p = 0; p < nparts; p++) {
            ProfScope ps(ctx, PROF_SPLIT);
            const u64 *a = (const u64 *)parts[p].A + kb * skA;
            const u64 *b = (const u64 *)parts[p].B + kb * skB;
            const u64 *b2 = parts[p].Bplus ? (const u64 *)parts[p].Bplus + kb * skB : nullptr;
            hipLaunchKernelGGL(split_limbs_kernel, dim3((u32)(lc / 64), (u32)(Mp / 64)), dim3(256), 0, ctx->stream, a,
                               (const u64 *)nullptr, r, klen, soA, skA, Apl, Mp, Kp, (u64)p * lc);
            hipLaunchKernelGGL(split_limbs_kernel, dim3((u32)(lc / 64), (u32)(Np / 64)), dim3(256), 0, ctx->stream, b,
                               b2, c, klen, soB, skB, Bpl, Np, Kp, (u64)p * lc);
        }
        {
            ProfScope ps(ctx, PROF_GEMM);
            if (te == TM)
                hipLaunchKernelGGL(limb_gemm256_kernel, dim3(tiles_m * tiles_n), dim3(512), 0, ctx->stream, Apl, Bpl,
                                   Cpl, (u32)Mp, (u32)Np, (u32)Kp, tiles_m, tiles_n);
            else
                hipLaunchKernelGGL(limb_gemm_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, ctx->stream, Apl, Bpl, Cpl,
                                   (u32)Mp, (u32)Np, (u32)Kp, tiles_m, tiles_n);
        }
        const u64 total = r * c;
        const u32 grid = (u32)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
        const bool last = kb + lc >= l;
        ProfScope pc(ctx, PROF_COMBINE);
        hipLaunchKernelGGL(combine_kernel, dim3(grid), dim3(256), 0, ctx->stream, Cpl, Mp, Np, r, c,
                           last ? (const u64 *)add : nullptr, last ? (const u64 *)sub : nullptr, (u64 *)Z,
                           layZ ? (u64)1 : c, layZ ? r : (u64)1, (accumulate || !first) ? 1 : 0);
        first = 0;
    }
    MPCX_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int mpcx_ring_gemm_u64(mpcx_ctx *ctx, const int64_t *A, const int64_t *B, int64_t *C, uint64_t r, uint64_t l,
                       uint64_t c, int layoutA, int layoutB, int layoutC, int accumulate) {
    MPCX_REQUIRE(A && B && C, "mpcx_ring_gemm_u64: NULL operand");
    GemmPart p{A, B, nullptr};
    return ring_gemm_parts(ctx, &p, 1, r, l, c, layoutA, layoutB, nullptr, nullptr, C, layoutC, accumulate);
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
