// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// A single-threaded CPU restatement of the per-party local arithmetic of the
// reference's ABY3Engine / SecurennEngine / MyEngine / PRgenerator (victerying/mpc).  It is
// written deliberately in the reference's own "one byte per bit", element-at-a-
// time style so that it is an independent check on the packed / fused HIP
// kernels.  Nothing in the product (mpc_amd/, include/) may call, link or load
// this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg do, and there only as the checker / the timed CPU baseline.
//
// Pinning status: the reference engines need Eigen 3.4.0, which is not vendored
// and not installed in the build image, so the engines cannot be compiled here
// without writing a stand-in header (not allowed).  This restatement is pinned
// by (a) the real reference PRgenerator compiled from its own sources into
// oracle/_ref (PRG words and generator state, any seed/length), (b) the PRG
// known-answer words recorded from the reference, and (c) the algebraic properties
// the reference's own tests assert (the reference holds no golden vectors of engine
// shares).  The bit patterns of engine shares are therefore "parity unpinned"
// against the reference: they are pinned against this restatement only.
//
// Every function cites the reference file:line it restates (paths relative to
// the reference root).
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

typedef uint64_t u64;
typedef int64_t i64;
typedef uint16_t u16;
typedef int8_t i8;
typedef uint8_t u8;

namespace {

// ---------------------------------------------------------------- PRG
// utils/PRgenerator.cpp:3-19
struct PrgState {
    u64 seed, lcg, hash;
};
inline u64 prv_step(PrgState &s) {
    s.seed *= s.lcg * 2 + 1;
    const u64 rs = s.seed >> 32 | s.seed << 32;
    s.hash += rs + 0xAAAAAAAAAAAAAAAAull;
    s.lcg += s.seed + 0x5555555555555555ull;
    s.seed ^= s.hash;
    return s.lcg ^ rs;
}

// The output sequence of PRgenerator(seed, lcg): raw outputs o[q], q = 0,1,...
//   q = 0..4        discarded by the constructor       (PRgenerator.cpp:23-26)
//   q = 5 + 2i      int_buff[i]                        (PRgenerator.cpp:29)
//   q = 6 + 2i      bool_buff[i] = o[q] % 2            (PRgenerator.cpp:30-31)
//   q = 320005 + k  k-th pop after construction        (PRgenerator.cpp:34-42, BUFF_NUM = 160000)
// "Extended buffer" semantics: int_buff[i] / bool_buff[i] are defined by the same
// formula for every i >= 0 (what the reference computes for i < BUFF_NUM).
// Stream index t = q - 5.
const u64 BUFF_NUM = 160000;
const u64 POP_T0 = 2 * BUFF_NUM;
const u64 MARK = 4096;

class Stream {
  public:
    Stream(u64 seed, u64 lcg) {
        PrgState s{seed, lcg, 0};
        for (int i = 0; i < 5; i++) prv_step(s);
        marks_.push_back(s);
    }
    // state before producing stream word t
    PrgState seek(u64 t) {
        u64 e = t / MARK;
        {
            std::lock_guard<std::mutex> g(mu_);
            while (marks_.size() <= e) {
                PrgState s = marks_.back();
                for (u64 i = 0; i < MARK; i++) prv_step(s);
                marks_.push_back(s);
            }
        }
        PrgState s;
        {
            std::lock_guard<std::mutex> g(mu_);
            s = marks_[e];
        }
        for (u64 i = e * MARK; i < t; i++) prv_step(s);
        return s;
    }

  private:
    std::mutex mu_;
    std::vector<PrgState> marks_;
};

std::mutex g_reg_mu;
std::map<std::pair<u64, u64>, std::unique_ptr<Stream>> g_streams;
Stream &stream(u64 seed, u64 lcg) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    auto key = std::make_pair(seed, lcg);
    auto it = g_streams.find(key);
    if (it == g_streams.end()) it = g_streams.emplace(key, std::make_unique<Stream>(seed, lcg)).first;
    return *it->second;
}

// sequential reader over the pop stream / the buffers of one generator
struct PopReader {
    PrgState s;
    PopReader(u64 seed, u64 pop_index) : s(stream(seed, 0).seek(POP_T0 + pop_index)) {}
    i64 pop() { return (i64)prv_step(s); }
};
struct BuffReader { // yields int_buff[i], bool_buff[i] for i = first, first+1, ...
    PrgState s;
    BuffReader(u64 seed, u64 first) : s(stream(seed, 0).seek(2 * first)) {}
    i64 next_int(i8 *b = nullptr) {
        i64 v = (i64)prv_step(s);
        u64 w = prv_step(s);
        if (b) *b = (i8)(w % 2);
        return v;
    }
};

// ---------------------------------------------------------------- Z_{2^64-1}
// mpc_engine/SecurennEngine.cpp:15-28
const u64 L_1 = ~0ull;
const u16 P = 67;
inline u64 add_L_1(u64 a, u64 b) {
    u64 c = a + b;
    bool carry = c < a;
    return (c + carry) % L_1;
}
inline u64 sub_L_1(u64 a, u64 b) {
    u64 _b = (L_1 - b) % L_1;
    return add_L_1(a, _b);
}

inline u64 ld(const i64 *m, u64 rows, u64 cols, int col_major, u64 i, u64 j) {
    return (u64)(col_major ? m[j * rows + i] : m[i * cols + j]);
}

// C (+)= A*B over Z_2^64, naive but cache-friendly (i,k,j order on row-major copies)
void ring_gemm(const i64 *A, const i64 *B, i64 *C, u64 r, u64 l, u64 c, int layA, int layB, int layC,
               int accumulate) {
    std::vector<u64> a(r * l), b(l * c), z(r * c, 0);
    for (u64 i = 0; i < r; i++)
        for (u64 k = 0; k < l; k++) a[i * l + k] = ld(A, r, l, layA, i, k);
    for (u64 k = 0; k < l; k++)
        for (u64 j = 0; j < c; j++) b[k * c + j] = ld(B, l, c, layB, k, j);
    for (u64 i = 0; i < r; i++)
        for (u64 k = 0; k < l; k++) {
            const u64 av = a[i * l + k];
            const u64 *bp = &b[k * c];
            u64 *zp = &z[i * c];
            for (u64 j = 0; j < c; j++) zp[j] += av * bp[j];
        }
    for (u64 i = 0; i < r; i++)
        for (u64 j = 0; j < c; j++) {
            u64 idx = layC ? j * r + i : i * c + j;
            C[idx] = (i64)((accumulate ? (u64)C[idx] : 0) + z[i * c + j]);
        }
}

// ---------------------------------------------------------------- ABY3 msb state
// mpc_engine/ABY3Engine.cpp:361-467, byte-per-bit exactly as the reference keeps it
struct MsbState {
    u64 n;
    std::array<std::vector<i8>, 2> G, Pv, backup, res; // res[0] = own product share of the current round
};

} // namespace

extern "C" {

// ================================================================ PRG
// utils/PRgenerator.cpp:21-33 (buffers, extended), :34-42 (pops)
int orc_prg_fill_buff(u64 seed, u64 lcg, u64 first, u64 count, i64 *ints, i8 *bools) {
    PrgState s = stream(seed, lcg).seek(2 * first);
    for (u64 i = 0; i < count; i++) {
        u64 a = prv_step(s), b = prv_step(s);
        if (ints) ints[i] = (i64)a;
        if (bools) bools[i] = (i8)(b % 2);
    }
    return 0;
}
int orc_prg_fill_pop(u64 seed, u64 lcg, u64 first_pop, u64 count, i64 *out) {
    PrgState s = stream(seed, lcg).seek(POP_T0 + first_pop);
    for (u64 i = 0; i < count; i++) out[i] = (i64)prv_step(s);
    return 0;
}
// (seed, lcg, hash) members of a PRgenerator that has popped `pop_index` words
int orc_prg_state_at_pop(u64 seed, u64 lcg, u64 pop_index, u64 *st) {
    PrgState s = stream(seed, lcg).seek(POP_T0 + pop_index);
    st[0] = s.seed;
    st[1] = s.lcg;
    st[2] = s.hash;
    return 0;
}
// FNV-1a-64 over 64-bit words: h = 0xCBF29CE484222325; h ^= w; h *= 0x100000001B3
u64 orc_fnv1a64_words(const u64 *w, u64 n) {
    u64 h = 0xCBF29CE484222325ull;
    for (u64 i = 0; i < n; i++) {
        h ^= w[i];
        h *= 0x100000001B3ull;
    }
    return h;
}

// ================================================================ ring helpers
int orc_ring_gemm_u64(const i64 *A, const i64 *B, i64 *C, u64 r, u64 l, u64 c, int layA, int layB, int layC,
                      int accumulate) {
    ring_gemm(A, B, C, r, l, c, layA, layB, layC, accumulate);
    return 0;
}
int orc_ring_add(const i64 *a, const i64 *b, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i64)((u64)a[i] + (u64)b[i]);
    return 0;
}
int orc_ring_sub(const i64 *a, const i64 *b, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i64)((u64)a[i] - (u64)b[i]);
    return 0;
}
// arithmetic shift right of each share (benchmark/benchmark1.cpp:106-112)
int orc_ring_sar(const i64 *a, i64 *out, u64 n, int k) {
    for (u64 i = 0; i < n; i++) out[i] = a[i] >> k;
    return 0;
}
// matrix layout change: src (rows x cols, layout lay_src) -> dst (layout lay_dst)
int orc_relayout(const i64 *src, i64 *dst, u64 rows, u64 cols, int lay_src, int lay_dst) {
    for (u64 i = 0; i < rows; i++)
        for (u64 j = 0; j < cols; j++) dst[lay_dst ? j * rows + i : i * cols + j] = (i64)ld(src, rows, cols, lay_src, i, j);
    return 0;
}

// ================================================================ ABY3
// share / share_matrix local part: mpc_engine/ABY3Engine.cpp:63-81, :83-106
// R[i] = pop(next) - pop(prev) in row-major element order; share0 = R (+ X on party 0,
// signalled by X != NULL).
int orc_aby3_share_local(const i64 *X, u64 n, u64 seed_next, u64 pop_next, u64 seed_prev, u64 pop_prev, i64 *out) {
    PopReader a(seed_next, pop_next), b(seed_prev, pop_prev);
    for (u64 i = 0; i < n; i++) {
        u64 r = (u64)a.pop() - (u64)b.pop();
        out[i] = (i64)(X ? r + (u64)X[i] : r);
    }
    return 0;
}
// share_bools local part: ABY3Engine.cpp:108-127
int orc_aby3_share_bools_local(const i8 *X, u64 n, u64 seed_next, u64 pop_next, u64 seed_prev, u64 pop_prev, i8 *out) {
    PopReader a(seed_next, pop_next), b(seed_prev, pop_prev);
    for (u64 i = 0; i < n; i++) {
        i8 r = (i8)(((u64)a.pop() % 2) ^ ((u64)b.pop() % 2));
        out[i] = X ? (i8)(r ^ X[i]) : r;
    }
    return 0;
}
// matrix_mul local part: ABY3Engine.cpp:189-197.  Inputs in `layout` (0 row-major,
// 1 column-major as Eigen stores MatrixXl); Z0 is written row-major = the wire
// order of matrix2string (utils/util.hpp:19-31) and the mask order of :190-197.
int orc_aby3_matmul_local(const i64 *X0, const i64 *X1, const i64 *Y0, const i64 *Y1, u64 r, u64 l, u64 c, int layout,
                          u64 seed_next, u64 pop_next, u64 seed_prev, u64 pop_prev, i64 *Z0) {
    ring_gemm(X0, Y0, Z0, r, l, c, layout, layout, 0, 0);
    ring_gemm(X0, Y1, Z0, r, l, c, layout, layout, 0, 1);
    ring_gemm(X1, Y0, Z0, r, l, c, layout, layout, 0, 1);
    PopReader a(seed_next, pop_next), b(seed_prev, pop_prev);
    for (u64 i = 0; i < r * c; i++) Z0[i] = (i64)((u64)Z0[i] + ((u64)a.pop() - (u64)b.pop()));
    return 0;
}
// vector_mul local part: ABY3Engine.cpp:220-225
int orc_aby3_vecmul_local(const i64 *X0, const i64 *X1, const i64 *Y0, const i64 *Y1, u64 n, u64 seed_next,
                          u64 pop_next, u64 seed_prev, u64 pop_prev, i64 *Z0) {
    PopReader a(seed_next, pop_next), b(seed_prev, pop_prev);
    for (u64 i = 0; i < n; i++) {
        u64 z = (u64)X0[i] * (u64)Y0[i] + (u64)X0[i] * (u64)Y1[i] + (u64)X1[i] * (u64)Y0[i];
        Z0[i] = (i64)(z + ((u64)a.pop() - (u64)b.pop()));
    }
    return 0;
}
// vecbool_mul local part: ABY3Engine.cpp:253-256 (one byte per boolean, no re-randomisation)
int orc_aby3_vecbool_mul_local(const i8 *X0, const i8 *X1, const i8 *Y0, const i8 *Y1, u64 n, i8 *Z0) {
    for (u64 i = 0; i < n; i++) Z0[i] = (i8)((X0[i] & Y0[i]) ^ (X1[i] & Y0[i]) ^ (X0[i] & Y1[i]));
    return 0;
}
// reveal / reveal_matrix: ABY3Engine.cpp:142,157 ; reveal_bools :171-174 ; reveal_bits :287-288
int orc_aby3_reveal3(const i64 *a, const i64 *b, const i64 *c, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i64)((u64)a[i] + (u64)b[i] + (u64)c[i]);
    return 0;
}
int orc_aby3_reveal3_bools(const i8 *a, const i8 *b, const i8 *c, i8 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i8)(a[i] ^ b[i] ^ c[i]);
    return 0;
}
int orc_aby3_reveal3_bits(const i64 *a, const i64 *b, const i64 *c, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = a[i] ^ b[i] ^ c[i];
    return 0;
}
// add_prepare: ABY3Engine.cpp:295-326
int orc_aby3_add_prepare_step1(const i64 *X0, const i64 *X1, u64 n, i64 *C0) {
    for (u64 i = 0; i < n; i++) C0[i] = X0[i] & X1[i];
    return 0;
}
int orc_aby3_add_prepare_step2(const i64 *X0, const i64 *X1, const i64 *C0, const i64 *C1, u64 n, i64 *P0, i64 *P1,
                               i64 *G0) {
    for (u64 i = 0; i < n; i++) {
        i64 c0 = (i64)((u64)C0[i] * 2), c1 = (i64)((u64)C1[i] * 2);
        P0[i] = c0 ^ X0[i];
        P1[i] = c1 ^ X1[i];
        G0[i] = (c0 & X0[i]) ^ (c0 & X1[i]) ^ (c1 & X0[i]);
    }
    return 0;
}
// msb: ABY3Engine.cpp:361-467.  The handle keeps the reference's byte-per-bit vectors.
void *orc_aby3_msb_begin(const i64 *G0, const i64 *G1, const i64 *P0, const i64 *P1, u64 n) {
    MsbState *st = new MsbState();
    st->n = n;
    const i64 *G[2] = {G0, G1}, *Pp[2] = {P0, P1};
    for (int i = 0; i < 2; i++) { // :370-389
        st->G[i].resize(n * 64);
        st->Pv[i].resize(n * 64);
        st->backup[i].resize(n);
        for (u64 k = 0; k < n; k++) {
            for (u64 b = 0; b < 64; b++) {
                st->G[i][k * 64 + b] = (i8)(((u64)G[i][k] >> b) & 1);
                st->Pv[i][k * 64 + b] = (i8)(((u64)Pp[i][k] >> b) & 1);
            }
            st->backup[i][k] = st->Pv[i][k * 64 + 63];
            st->G[i][k * 64 + 63] = 0;
            st->Pv[i][k * 64 + 63] = 1;
        }
    }
    return st;
}
// one round, before the messenger call: builds op1/op2 (:397-413) and the local part of
// vecbool_mul (:253-256).  msg_out = result[0], 2*temp bytes per element.
int orc_aby3_msb_round_pre(void *h, u64 temp, i8 *msg_out) {
    MsbState *st = (MsbState *)h;
    u64 n = st->n;
    std::array<std::vector<i8>, 2> op1, op2;
    for (int i = 0; i < 2; i++) {
        op1[i].resize(n * 2 * temp);
        op2[i].resize(n * 2 * temp);
        for (u64 k = 0; k < n; k++)
            for (u64 m = 0; m < temp; m++) {
                op1[i][k * 2 * temp + 2 * m] = st->Pv[i][k * 2 * temp + 2 * m + 1];
                op2[i][k * 2 * temp + 2 * m] = st->G[i][k * 2 * temp + 2 * m];
                op1[i][k * 2 * temp + 2 * m + 1] = st->Pv[i][k * 2 * temp + 2 * m + 1];
                op2[i][k * 2 * temp + 2 * m + 1] = st->Pv[i][k * 2 * temp + 2 * m];
            }
    }
    st->res[0].resize(n * 2 * temp);
    for (u64 i = 0; i < n * 2 * temp; i++) {
        st->res[0][i] = (i8)((op1[0][i] & op2[0][i]) ^ (op1[1][i] & op2[0][i]) ^ (op1[0][i] & op2[1][i]));
        msg_out[i] = st->res[0][i];
    }
    return 0;
}
// after the messenger call: result[1] = msg_in, then :416-428 and :455-456
int orc_aby3_msb_round_post(void *h, u64 temp, const i8 *msg_in) {
    MsbState *st = (MsbState *)h;
    u64 n = st->n;
    st->res[1].assign(msg_in, msg_in + n * 2 * temp);
    for (int i = 0; i < 2; i++) {
        std::vector<i8> Gn(n * temp), Pn(n * temp);
        for (u64 k = 0; k < n; k++)
            for (u64 m = 0; m < temp; m++) {
                Gn[k * temp + m] = (i8)(st->res[i][k * 2 * temp + 2 * m] ^ st->G[i][k * 2 * temp + 2 * m + 1]);
                Pn[k * temp + m] = st->res[i][k * 2 * temp + 2 * m + 1];
            }
        st->G[i].swap(Gn);
        st->Pv[i].swap(Pn);
    }
    return 0;
}
int orc_aby3_msb_final(void *h, i8 *Z0, i8 *Z1) { // :459-466
    MsbState *st = (MsbState *)h;
    for (u64 k = 0; k < st->n; k++) {
        Z0[k] = (i8)(st->G[0][k] ^ st->backup[0][k]);
        Z1[k] = (i8)(st->G[1][k] ^ st->backup[1][k]);
    }
    delete st;
    return 0;
}
// trunc local part: ABY3Engine.cpp:480-503
int orc_aby3_trunc_local(int party, const i64 *X0, const i64 *X1, u64 n, int number, u64 seed12, u64 pop12, i64 *Y0) {
    if (party == 0) {
        for (u64 i = 0; i < n; i++) Y0[i] = X0[i] >> number;
    } else if (party == 1) {
        PopReader p(seed12, pop12);
        for (u64 i = 0; i < n; i++) {
            i64 t = (i64)((u64)X0[i] + (u64)X1[i]);
            t = t >> number;
            Y0[i] = (i64)((u64)t - (u64)p.pop());
        }
    } else {
        PopReader p(seed12, pop12);
        for (u64 i = 0; i < n; i++) Y0[i] = p.pop();
    }
    return 0;
}

// bit_inject, party 0, before the round: ABY3Engine.cpp:533-620.
// s01 pops from pop01: C01[n], (W01_0,W01_1)[n] interleaved, C10[n], (W10_0,W10_1)[n] interleaved  (6n words)
// s20 pops from pop20: C02[n], (W02_0,W02_1)[n] interleaved                                        (3n words)
// to_p1 = (M0_0^W02_0)[n] | (M0_1^W02_1)[n];  to_p2 = (M0_0^W01_0)[n] | (M0_1^W01_1)[n] | W10_{B0}[n]
int orc_aby3_bit_inject_p0_pre(const i64 *A0, const i8 *B0, const i8 *B1, u64 n, u64 seed01, u64 pop01, u64 seed20,
                               u64 pop20, i64 *to_p1, i64 *to_p2) {
    std::vector<i64> s01(6 * n), s20(3 * n);
    orc_prg_fill_pop(seed01, 0, pop01, 6 * n, s01.data());
    orc_prg_fill_pop(seed20, 0, pop20, 3 * n, s20.data());
    for (u64 i = 0; i < n; i++) {
        u64 C01 = s01[i], C02 = s20[i];
        u64 M0 = (u64)(i64)(0 ^ B0[i] ^ B1[i]) * (u64)A0[i] - C01 - C02;
        u64 M1 = (u64)(i64)(1 ^ B0[i] ^ B1[i]) * (u64)A0[i] - C01 - C02;
        u64 W02_0 = s20[n + 2 * i], W02_1 = s20[n + 2 * i + 1];
        u64 W01_0 = s01[n + 2 * i], W01_1 = s01[n + 2 * i + 1];
        u64 W10_0 = s01[4 * n + 2 * i], W10_1 = s01[4 * n + 2 * i + 1];
        to_p1[i] = (i64)(M0 ^ W02_0);
        to_p1[n + i] = (i64)(M1 ^ W02_1);
        to_p2[i] = (i64)(M0 ^ W01_0);
        to_p2[n + i] = (i64)(M1 ^ W01_1);
        to_p2[2 * n + i] = (i64)(B0[i] == 1 ? W10_1 : W10_0);
    }
    return 0;
}
// party 0, after the round: ABY3Engine.cpp:622-682.  from_p1 = (M1_0^W12_0)|(M1_1^W12_1) (2n), from_p2 = W12_{b} (n)
int orc_aby3_bit_inject_p0_post(const i8 *B0, u64 n, u64 seed01, u64 pop01, u64 seed20, u64 pop20, const i64 *from_p1,
                                const i64 *from_p2, i64 *C0, i64 *C1) {
    std::vector<i64> s01(6 * n), s20(3 * n);
    orc_prg_fill_pop(seed01, 0, pop01, 6 * n, s01.data());
    orc_prg_fill_pop(seed20, 0, pop20, 3 * n, s20.data());
    for (u64 i = 0; i < n; i++) {
        u64 C01 = s01[i], C02 = s20[i], C10 = s01[3 * n + i];
        u64 A12_0 = (u64)(B0[i] == 1 ? from_p1[n + i] : from_p1[i]) ^ (u64)from_p2[i];
        C0[i] = (i64)(C02 + A12_0); // A0j[0] + A12j[0]
        C1[i] = (i64)(C01 + C10);   // A0j[1] + A12j[1]
    }
    return 0;
}
// party 1, before: ABY3Engine.cpp:702-790.
// s01 (prgs[0]) pops: C01[n], (W01_0,W01_1)[n], C10[n], (W10_0,W10_1)[n]; s12 (prgs[2]) pops: C12[n], (W12_0,W12_1)[n]
// to_p2 = W01_{B1}[n] | (M1_0^W10_0)[n] | (M1_1^W10_1)[n];  to_p0 = (M1_0^W12_0)[n] | (M1_1^W12_1)[n]
int orc_aby3_bit_inject_p1_pre(const i64 *A0, const i64 *A1, const i8 *B0, const i8 *B1, u64 n, u64 seed01, u64 pop01,
                               u64 seed12, u64 pop12, i64 *to_p0, i64 *to_p2) {
    std::vector<i64> s01(6 * n), s12(3 * n);
    orc_prg_fill_pop(seed01, 0, pop01, 6 * n, s01.data());
    orc_prg_fill_pop(seed12, 0, pop12, 3 * n, s12.data());
    for (u64 i = 0; i < n; i++) {
        u64 W01_0 = s01[n + 2 * i], W01_1 = s01[n + 2 * i + 1];
        u64 C10 = s01[3 * n + i], C12 = s12[i];
        u64 a = (u64)A0[i] + (u64)A1[i];
        u64 M0 = (u64)(i64)(0 ^ B0[i] ^ B1[i]) * a - C10 - C12;
        u64 M1 = (u64)(i64)(1 ^ B0[i] ^ B1[i]) * a - C10 - C12;
        u64 W10_0 = s01[4 * n + 2 * i], W10_1 = s01[4 * n + 2 * i + 1];
        u64 W12_0 = s12[n + 2 * i], W12_1 = s12[n + 2 * i + 1];
        to_p2[i] = (i64)(B1[i] == 1 ? W01_1 : W01_0);
        to_p2[n + i] = (i64)(M0 ^ W10_0);
        to_p2[2 * n + i] = (i64)(M1 ^ W10_1);
        to_p0[i] = (i64)(M0 ^ W12_0);
        to_p0[n + i] = (i64)(M1 ^ W12_1);
    }
    return 0;
}
// party 1, after: ABY3Engine.cpp:791-849.  from_p0 = (M0_0^W02_0)|(M0_1^W02_1), from_p2 = W02_{b}
int orc_aby3_bit_inject_p1_post(const i8 *B1, u64 n, u64 seed01, u64 pop01, u64 seed12, u64 pop12, const i64 *from_p0,
                                const i64 *from_p2, i64 *C0, i64 *C1) {
    std::vector<i64> s01(6 * n), s12(3 * n);
    orc_prg_fill_pop(seed01, 0, pop01, 6 * n, s01.data());
    orc_prg_fill_pop(seed12, 0, pop12, 3 * n, s12.data());
    for (u64 i = 0; i < n; i++) {
        u64 C01 = s01[i], C10 = s01[3 * n + i], C12 = s12[i];
        u64 A0_1 = (u64)(B1[i] == 1 ? from_p0[n + i] : from_p0[i]) ^ (u64)from_p2[i];
        C0[i] = (i64)(C01 + C10);  // A0j[0] + A12j[0]
        C1[i] = (i64)(A0_1 + C12); // A0j[1] + A12j[1]
    }
    return 0;
}
// party 2, before: ABY3Engine.cpp:869-939.  s20 (prgs[0]): C02[n], (W02_0,W02_1)[n]; s12 (prgs[1]): C12[n], (W12_0,W12_1)[n]
// to_p1 = W02_{B0}[n];  to_p0 = W12_{B1}[n]
int orc_aby3_bit_inject_p2_pre(const i8 *B0, const i8 *B1, u64 n, u64 seed20, u64 pop20, u64 seed12, u64 pop12,
                               i64 *to_p0, i64 *to_p1) {
    std::vector<i64> s20(3 * n), s12(3 * n);
    orc_prg_fill_pop(seed20, 0, pop20, 3 * n, s20.data());
    orc_prg_fill_pop(seed12, 0, pop12, 3 * n, s12.data());
    for (u64 i = 0; i < n; i++) {
        to_p1[i] = B0[i] == 1 ? s20[n + 2 * i + 1] : s20[n + 2 * i];
        to_p0[i] = B1[i] == 1 ? s12[n + 2 * i + 1] : s12[n + 2 * i];
    }
    return 0;
}
// party 2, after: ABY3Engine.cpp:940-1017.
// from_p0 = (M0_0^W01_0)|(M0_1^W01_1)|W10_{b} (3n), from_p1 = W01_{b}|(M1_0^W10_0)|(M1_1^W10_1) (3n)
int orc_aby3_bit_inject_p2_post(const i8 *B0, const i8 *B1, u64 n, u64 seed20, u64 pop20, u64 seed12, u64 pop12,
                                const i64 *from_p0, const i64 *from_p1, i64 *C0, i64 *C1) {
    std::vector<i64> s20(3 * n), s12(3 * n);
    orc_prg_fill_pop(seed20, 0, pop20, 3 * n, s20.data());
    orc_prg_fill_pop(seed12, 0, pop12, 3 * n, s12.data());
    for (u64 i = 0; i < n; i++) {
        u64 C02 = s20[i], C12 = s12[i];
        u64 A0_0 = (u64)(B0[i] == 1 ? from_p0[n + i] : from_p0[i]) ^ (u64)from_p1[i];
        u64 A12_1 = (u64)(B1[i] == 1 ? from_p1[2 * n + i] : from_p1[n + i]) ^ (u64)from_p0[2 * n + i];
        C0[i] = (i64)(A0_0 + C12);  // A0j[0] + A12j[0]
        C1[i] = (i64)(C02 + A12_1); // A0j[1] + A12j[1]
    }
    return 0;
}

// ================================================================ SecureNN
// share: mpc_engine/SecurennEngine.cpp:318-334 (party 0: -pop(s01); party 1: X + pop(s01); party 2: zeros)
int orc_snn_share_local(int party, const i64 *X, u64 n, u64 seed01, u64 pop01, i64 *out) {
    if (party == 2) {
        for (u64 i = 0; i < n; i++) out[i] = 0;
        return 0;
    }
    PopReader p(seed01, pop01);
    for (u64 i = 0; i < n; i++) {
        u64 t = (u64)p.pop();
        out[i] = (i64)(party == 0 ? (u64)0 - t : (u64)X[i] + t);
    }
    return 0;
}
// reveal_L_1: SecurennEngine.cpp:448,459
int orc_snn_add_L1(const i64 *a, const i64 *b, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i64)add_L_1((u64)a[i], (u64)b[i]);
    return 0;
}
// reveal_zp: SecurennEngine.cpp:482,496
int orc_snn_add_zp(const u16 *a, const u16 *b, u16 *out, u64 n64) {
    for (u64 i = 0; i < n64; i++) {
        u16 t = (u16)(a[i] + b[i]);
        out[i] = t % P;
    }
    return 0;
}
// matrix_mul, party 2 (dealer): SecurennEngine.cpp:68-117.  bufS = int_buff of the self stream.
// msg_i = A_i | B_i | C_i, each row-major.
int orc_snn_beaver_dealer_matmul(u64 seedS, u64 r, u64 l, u64 c, i64 *msg0, i64 *msg1) {
    i64 *msg[2] = {msg0, msg1};
    u64 na = r * l, nb = l * c, nc = r * c;
    orc_prg_fill_buff(seedS, 0, 0, na, msg0, nullptr);
    orc_prg_fill_buff(seedS, 0, na, na, msg1, nullptr);
    orc_prg_fill_buff(seedS, 0, 2 * na, nb, msg0 + na, nullptr);
    orc_prg_fill_buff(seedS, 0, 2 * na + nb, nb, msg1 + na, nullptr);
    orc_prg_fill_buff(seedS, 0, 2 * na + 2 * nb, nc, msg0 + na + nb, nullptr);
    std::vector<i64> A(na), B(nb), Cp(nc);
    orc_ring_add(msg[0], msg[1], A.data(), na);
    orc_ring_add(msg[0] + na, msg[1] + na, B.data(), nb);
    ring_gemm(A.data(), B.data(), Cp.data(), r, l, c, 0, 0, 0, 0);
    orc_ring_sub(Cp.data(), msg0 + na + nb, msg1 + na + nb, nc);
    return 0;
}
// party 0/1 after round 1: E = X - A, F = Y - B (:150-155).  ef_out = E | F row-major.
int orc_snn_beaver_party_open_matmul(const i64 *X, const i64 *Y, int layout, const i64 *triple, u64 r, u64 l, u64 c,
                                     i64 *ef_out) {
    for (u64 i = 0; i < r; i++)
        for (u64 k = 0; k < l; k++) ef_out[i * l + k] = (i64)(ld(X, r, l, layout, i, k) - (u64)triple[i * l + k]);
    for (u64 k = 0; k < l; k++)
        for (u64 j = 0; j < c; j++)
            ef_out[r * l + k * c + j] = (i64)(ld(Y, l, c, layout, k, j) - (u64)triple[r * l + k * c + j]);
    return 0;
}
// party 0/1 after round 2 (:159-167): Z = C + E*B + A*F + E*F*party ; Z written in `layoutZ`
int orc_snn_beaver_party_finish_matmul(int party, const i64 *triple, const i64 *ef_own, const i64 *ef_peer, u64 r,
                                       u64 l, u64 c, int layoutZ, i64 *Z) {
    u64 na = r * l, nb = l * c, nc = r * c;
    std::vector<i64> E(na), F(nb), z(nc);
    orc_ring_add(ef_own, ef_peer, E.data(), na);
    orc_ring_add(ef_own + na, ef_peer + na, F.data(), nb);
    const i64 *A = triple, *B = triple + na, *C = triple + na + nb;
    ring_gemm(E.data(), B, z.data(), r, l, c, 0, 0, 0, 0);
    ring_gemm(A, F.data(), z.data(), r, l, c, 0, 0, 0, 1);
    if (party == 1) ring_gemm(E.data(), F.data(), z.data(), r, l, c, 0, 0, 0, 1);
    for (u64 i = 0; i < nc; i++) z[i] = (i64)((u64)z[i] + (u64)C[i]);
    orc_relayout(z.data(), Z, r, c, 0, layoutZ);
    return 0;
}
// vector_mul, party 2: SecurennEngine.cpp:176-216.  A_i[j]=buf[3(i n + j)], B=+1, C_0=+2;
// msg_i = A_i | B_i | C_i.  (n_total, i0): this call covers elements [i0, i0+n) of a vector of n_total.
int orc_snn_beaver_dealer_vecmul(u64 seedS, u64 n_total, u64 i0, u64 n, i64 *A0, i64 *B0, i64 *C0, i64 *A1, i64 *B1,
                                 i64 *C1) {
    std::vector<i64> w0(3 * n), w1(3 * n);
    orc_prg_fill_buff(seedS, 0, 3 * i0, 3 * n, w0.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, 3 * (n_total + i0), 3 * n, w1.data(), nullptr);
    for (u64 j = 0; j < n; j++) {
        A0[j] = w0[3 * j];
        B0[j] = w0[3 * j + 1];
        C0[j] = w0[3 * j + 2];
        A1[j] = w1[3 * j];
        B1[j] = w1[3 * j + 1];
        C1[j] = (i64)(((u64)A0[j] + (u64)A1[j]) * ((u64)B0[j] + (u64)B1[j]) - (u64)C0[j]);
    }
    return 0;
}
// party 0/1 (:268-276)
int orc_snn_beaver_party_open_vecmul(const i64 *X, const i64 *Y, const i64 *A, const i64 *B, u64 n, i64 *E, i64 *F) {
    for (u64 i = 0; i < n; i++) {
        E[i] = (i64)((u64)X[i] - (u64)A[i]);
        F[i] = (i64)((u64)Y[i] - (u64)B[i]);
    }
    return 0;
}
// (:287-295)
int orc_snn_beaver_party_finish_vecmul(int party, const i64 *A, const i64 *B, const i64 *C, const i64 *E_own,
                                       const i64 *F_own, const i64 *E_peer, const i64 *F_peer, u64 n, i64 *Z) {
    for (u64 i = 0; i < n; i++) {
        u64 E = (u64)E_own[i] + (u64)E_peer[i], F = (u64)F_own[i] + (u64)F_peer[i];
        Z[i] = (i64)((u64)C[i] + E * (u64)B[i] + F * (u64)A[i] + F * E * (u64)party);
    }
    return 0;
}

// privateCompare, party 0/1: SecurennEngine.cpp:513-595.
// s_k = (u16)buf01[64 i + k] % 67 (0 -> 1), u_k = (u16)buf01[64 n_total + 64 i + k] % 67 (0 -> 1), i global.
int orc_snn_pc_party(int party, const u16 *xbits, const i8 *beta, const i64 *r, u64 n_total, u64 i0, u64 n, u64 seed01,
                     u16 *d_out) {
    BuffReader rs(seed01, 64 * i0), ru(seed01, 64 * n_total + 64 * i0);
    const u64 party_id = (u64)party;
    for (u64 i = 0; i < n; i++) {
        u16 S[64], U[64], C[64], W[64], R[64], T[64];
        for (int k = 0; k < 64; k++) {
            S[k] = (u16)rs.next_int();
            U[k] = (u16)ru.next_int();
            S[k] = S[k] % P;
            if (0 == S[k]) S[k] = 1;
            U[k] = U[k] % P;
            if (0 == U[k]) U[k] = 1;
        }
        const u16 *X = xbits + 64 * i;
        for (int k = 0; k < 64; k++) {
            R[k] = (u16)(((u64)r[i] >> k) & 1);
            T[k] = (u16)((((u64)r[i] + 1) >> k) & 1);
        }
        u16 sum_w = 0;
        for (int k = 63; k != -1; k--) {
            if (beta[i] == 0) {
                C[k] = (u16)((X[k] * (P - 1)) % P + (u16)party_id * R[k] + (u16)party_id + sum_w);
                C[k] = C[k] % P;
                W[k] = (u16)(X[k] + (u16)party_id * R[k] + (P - 2) * R[k] * X[k]);
                W[k] = W[k] % P;
                sum_w = (u16)(sum_w + W[k]);
            } else if (beta[i] == 1 && r[i] != -1) {
                C[k] = (u16)((u64)(P - 1) * party_id * T[k] + X[k] + (u16)party_id + sum_w);
                C[k] = C[k] % P;
                W[k] = (u16)(X[k] + (u16)party_id * T[k] + (P - 2) * T[k] * X[k]);
                W[k] = W[k] % P;
                sum_w = (u16)(sum_w + W[k]);
            } else {
                if (k != 0) {
                    C[k] = (u16)(((u64)1 - party_id) * (u64)(1 + U[k]) + ((u64)P - party_id) * U[k]);
                    C[k] = C[k] % P;
                } else {
                    C[k] = party_id == 0 ? U[k] : (u16)(P - U[k]);
                }
            }
        }
        for (int k = 0; k < 64; k++) {
            u16 d = (u16)(S[k] * C[k]);
            d_out[64 * i + k] = d % P;
        }
    }
    return 0;
}
// privateCompare, party 2: SecurennEngine.cpp:655-669
int orc_snn_pc_helper(const u16 *d0, const u16 *d1, u64 n, i8 *beta_out) {
    for (u64 i = 0; i < n; i++) {
        beta_out[i] = 0;
        for (int k = 0; k < 64; k++) {
            u16 d = (u16)(d0[64 * i + k] + d1[64 * i + k]);
            d = d % P;
            if (0 == d) {
                beta_out[i] = 1;
                break;
            }
        }
    }
    return 0;
}

// shareConvert, party 0/1, before round 1 (:692-722): x_j = a_j + r_j
int orc_snn_sc_party_open(int party, const i64 *A, u64 n_total, u64 i0, u64 n, u64 seed01, i64 *x_out) {
    std::vector<i64> r0(n), r1(n);
    orc_prg_fill_buff(seed01, 0, i0, n, r0.data(), nullptr);
    orc_prg_fill_buff(seed01, 0, n_total + i0, n, r1.data(), nullptr);
    for (u64 i = 0; i < n; i++) {
        u64 R = (u64)r0[i] + (u64)r1[i];
        u64 rs1 = (u64)r1[i];
        if (R == 0) rs1 += 1;
        x_out[i] = (i64)((u64)A[i] + (party == 0 ? (u64)r0[i] : rs1));
    }
    return 0;
}
// the plaintext inputs the party feeds privateCompare in round 3 (:735-740): eta'' and r-1
int orc_snn_sc_party_pc_inputs(u64 n_total, u64 i0, u64 n, u64 seed01, i8 *eta2_out, i64 *rm1_out) {
    std::vector<i64> r0(n), r1(n);
    orc_prg_fill_buff(seed01, 0, i0, n, r0.data(), eta2_out);
    orc_prg_fill_buff(seed01, 0, n_total + i0, n, r1.data(), nullptr);
    for (u64 i = 0; i < n; i++) {
        u64 R = (u64)r0[i] + (u64)r1[i];
        if (R == 0) R += 1;
        rm1_out[i] = (i64)(R - 1);
    }
    return 0;
}
// after round 4 (:744-766): y_j = a_j -L theta_j
int orc_snn_sc_party_finish(int party, const i64 *A, const i64 *delta_j, const i64 *eta1_j, u64 n_total, u64 i0, u64 n,
                            u64 seed01, i64 *y_out) {
    std::vector<i64> r0(n), r1(n);
    std::vector<i8> eta2(n);
    orc_prg_fill_buff(seed01, 0, i0, n, r0.data(), eta2.data());
    orc_prg_fill_buff(seed01, 0, n_total + i0, n, r1.data(), nullptr);
    const u64 party_id = (u64)party;
    for (u64 i = 0; i < n; i++) {
        u64 R = (u64)r0[i] + (u64)r1[i];
        u64 rs[2] = {(u64)r0[i], (u64)r1[i]};
        if (R == 0) {
            R += 1;
            rs[1] += 1;
        }
        i8 alpha = R < rs[0];
        u64 x = (u64)A[i] + rs[party];
        i8 beta = x < rs[party];
        u64 eta = add_L_1((u64)eta1_j[i], party_id * (u64)(i64)eta2[i]);
        u64 temp = add_L_1((u64)eta1_j[i], (u64)eta1_j[i]);
        temp = temp * (u64)(i64)eta2[i];
        eta = sub_L_1(eta, temp);
        u64 tempj = (1 - party_id) * (L_1 - 1 - (u64)(i64)alpha);
        u64 theta = add_L_1((u64)(i64)beta, tempj);
        theta = add_L_1(theta, eta);
        theta = add_L_1(theta, (u64)delta_j[i]);
        y_out[i] = (i64)sub_L_1((u64)A[i], theta);
    }
    return 0;
}
// shareConvert, party 2, after round 1 (:816-859): message parts for P0 and P1
int orc_snn_sc_helper_split(const i64 *x0, const i64 *x1, u64 n_total, u64 i0, u64 n, u64 seedS, u16 *xbits0,
                            i64 *delta0, u16 *xbits1, i64 *delta1) {
    BuffReader rb(seedS, 64 * i0);
    std::vector<i64> d0(n);
    orc_prg_fill_buff(seedS, 0, i0, n, d0.data(), nullptr);
    (void)n_total;
    for (u64 i = 0; i < n; i++) {
        u64 x = (u64)x0[i] + (u64)x1[i];
        u64 delta = x < (u64)x0[i];
        for (int k = 0; k < 64; k++) {
            u16 bit = (u16)((x >> k) & 1);
            u16 s0 = (u16)rb.next_int();
            s0 = s0 % P;
            u16 s1 = (u16)(bit + P - s0);
            s1 = s1 % P;
            xbits0[64 * i + k] = s0;
            xbits1[64 * i + k] = s1;
        }
        u64 D0 = (u64)d0[i] % L_1;
        delta0[i] = (i64)D0;
        delta1[i] = (i64)sub_L_1(delta, D0);
    }
    return 0;
}
// party 2 after round 3 (:865-874): shares of eta' over Z_{L-1}
int orc_snn_sc_helper_eta(const i8 *eta1_plain, u64 n_total, u64 i0, u64 n, u64 seedS, i64 *eta0, i64 *eta1) {
    std::vector<i64> w(n);
    orc_prg_fill_buff(seedS, 0, n_total + i0, n, w.data(), nullptr);
    for (u64 i = 0; i < n; i++) {
        u64 e0 = (u64)w[i] % L_1;
        eta0[i] = (i64)e0;
        eta1[i] = (i64)sub_L_1((u64)(i64)eta1_plain[i], e0);
    }
    return 0;
}

// msb_L_1, party 2, before round 1 (:905-940)
int orc_snn_msbl1_helper_deal(u64 n_total, u64 i0, u64 n, u64 seedS, i64 *x0, i64 *xlsb0, u16 *xbits0, i64 *x1,
                              i64 *xlsb1, u16 *xbits1) {
    std::vector<i64> w0(n), w1(n), w2(n);
    orc_prg_fill_buff(seedS, 0, i0, n, w0.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, n_total + i0, n, w1.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, 2 * n_total + i0, n, w2.data(), nullptr);
    BuffReader rb(seedS, 3 * n_total + 64 * i0);
    for (u64 i = 0; i < n; i++) {
        u64 X0 = (u64)w0[i] % L_1, X1 = (u64)w1[i] % L_1;
        u64 X = add_L_1(X0, X1);
        x0[i] = (i64)X0;
        x1[i] = (i64)X1;
        xlsb0[i] = w2[i];
        xlsb1[i] = (i64)((u64)(X & 1) - (u64)w2[i]);
        for (int k = 0; k < 64; k++) {
            u16 bit = (u16)((X >> k) & 1);
            u16 s1 = (u16)rb.next_int();
            s1 = s1 % P;
            u16 s0 = (u16)(bit + P - s1);
            s0 = s0 % P;
            xbits0[64 * i + k] = s0;
            xbits1[64 * i + k] = s1;
        }
    }
    return 0;
}
// party 2 after round 3 (:950-957)
int orc_snn_msbl1_helper_beta(const i8 *beta1_plain, u64 n_total, u64 i0, u64 n, u64 seedS, i64 *b0, i64 *b1) {
    std::vector<i64> w(n);
    orc_prg_fill_buff(seedS, 0, 4 * n_total + i0, n, w.data(), nullptr);
    for (u64 i = 0; i < n; i++) {
        b0[i] = w[i];
        b1[i] = (i64)((u64)(i64)beta1_plain[i] - (u64)w[i]);
    }
    return 0;
}
// party 0/1 before round 2 (:1002-1007): r_j = 2a +L x_j
int orc_snn_msbl1_party_open(const i64 *A, const i64 *xj, u64 n, i64 *r_out) {
    for (u64 i = 0; i < n; i++) {
        u64 y = add_L_1((u64)A[i], (u64)A[i]);
        r_out[i] = (i64)add_L_1(y, (u64)xj[i]);
    }
    return 0;
}
// plaintext privateCompare inputs of round 3 (:985, :1011-1016): beta = boolbuf01[i], R = r_peer +L r_own
int orc_snn_msbl1_party_pc_inputs(const i64 *r_own, const i64 *r_peer, u64 i0, u64 n, u64 seed01, i8 *beta_out,
                                  i64 *R_out) {
    std::vector<i64> dummy(n);
    orc_prg_fill_buff(seed01, 0, i0, n, dummy.data(), beta_out);
    for (u64 i = 0; i < n; i++) R_out[i] = (i64)add_L_1((u64)r_peer[i], (u64)r_own[i]);
    return 0;
}
// after round 4 (:1022-1032): gamma_j (the reference's Beta_j) and delta_j
int orc_snn_msbl1_party_mid(int party, const i64 *beta1_j, const i64 *xlsb_j, const i8 *beta0, const i64 *R, u64 n,
                            i64 *gamma_out, i64 *delta_out) {
    const u64 pid = (u64)party;
    for (u64 i = 0; i < n; i++) {
        u64 r0 = (u64)R[i] % 2;
        u64 b0 = (u64)(i64)beta0[i];
        gamma_out[i] = (i64)((u64)beta1_j[i] + pid * b0 - 2 * b0 * (u64)beta1_j[i]);
        delta_out[i] = (i64)((u64)xlsb_j[i] + pid * r0 - 2 * r0 * (u64)xlsb_j[i]);
    }
    return 0;
}
// after vector_mul (:1034-1037)
int orc_snn_msbl1_party_finish(const i64 *gamma, const i64 *delta, const i64 *theta, u64 n, i64 *B) {
    for (u64 i = 0; i < n; i++) B[i] = (i64)((u64)gamma[i] + (u64)delta[i] - 2 * (u64)theta[i]);
    return 0;
}
// msb entry (:1094-1098): 2a
int orc_ring_shl1(const i64 *a, i64 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i64)((u64)a[i] << 1);
    return 0;
}

// ================================================================ MyEngine
// mpc_engine/MyEngine.cpp: two-party additive shares (parties 0/1) with party 2 as the dealer; every product,
// msb and bit_inject costs few rounds because the correlated randomness comes from the *buffers* of the
// pairwise streams: party 2's prgs[0] / prgs[1] are the streams it shares with party 0 / 1 (KEY20 / KEY12,
// :21-24) and party j's prgs[2] is that same stream (:11-19), so both sides read identical int_buff words.
// share / share_matrix (:54-89) are orc_snn_share_local; reveal* (:91-143, :777-877) are orc_ring_add,
// orc_snn_add_L1, orc_snn_add_zp and a byte XOR.

// matrix_mul, party 2 (:152-184): A_j = buf_j[k*l + ..], B_j = buf_j[r*l + ..], C_0 = bufS[i*c + k],
// C_1 = (A_0 + A_1)(B_0 + B_1) - C_0; C_j (row-major) is the message to party j.
int orc_my_beaver_dealer_matmul(u64 seed0, u64 seed1, u64 seedS, u64 r, u64 l, u64 c, i64 *C0, i64 *C1) {
    const u64 na = r * l, nb = l * c, nc = r * c;
    std::vector<i64> A0(na), A1(na), B0(nb), B1(nb), As(na), Bs(nb), Cs(nc);
    orc_prg_fill_buff(seed0, 0, 0, na, A0.data(), nullptr);
    orc_prg_fill_buff(seed1, 0, 0, na, A1.data(), nullptr);
    orc_prg_fill_buff(seed0, 0, na, nb, B0.data(), nullptr);
    orc_prg_fill_buff(seed1, 0, na, nb, B1.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, 0, nc, C0, nullptr);
    for (u64 i = 0; i < na; i++) As[i] = (i64)((u64)A0[i] + (u64)A1[i]);
    for (u64 i = 0; i < nb; i++) Bs[i] = (i64)((u64)B0[i] + (u64)B1[i]);
    ring_gemm(As.data(), Bs.data(), Cs.data(), r, l, c, 0, 0, 0, 0);
    for (u64 i = 0; i < nc; i++) C1[i] = (i64)((u64)Cs[i] - (u64)C0[i]);
    return 0;
}
// party 0/1 before the round (:187-205): A_j = buf2[k*l + ..], B_j = buf2[r*l + ..]; ef_out = (X - A) | (Y - B), row-major
int orc_my_beaver_party_open_matmul(const i64 *X, const i64 *Y, int layout, u64 seed2, u64 r, u64 l, u64 c, i64 *ef_out) {
    const u64 na = r * l, nb = l * c;
    std::vector<i64> A(na), B(nb);
    orc_prg_fill_buff(seed2, 0, 0, na, A.data(), nullptr);
    orc_prg_fill_buff(seed2, 0, na, nb, B.data(), nullptr);
    for (u64 i = 0; i < r; i++)
        for (u64 k = 0; k < l; k++) ef_out[i * l + k] = (i64)(ld(X, r, l, layout, i, k) - (u64)A[i * l + k]);
    for (u64 i = 0; i < l; i++)
        for (u64 k = 0; k < c; k++) ef_out[na + i * c + k] = (i64)(ld(Y, l, c, layout, i, k) - (u64)B[i * c + k]);
    return 0;
}
// party 0/1 after the round (:208-217): Z = C + E*B + A*F + E*F*party
int orc_my_beaver_party_finish_matmul(int party, u64 seed2, const i64 *C, const i64 *ef_own, const i64 *ef_peer, u64 r,
                                      u64 l, u64 c, int layoutZ, i64 *Z) {
    const u64 na = r * l, nb = l * c, nc = r * c;
    std::vector<i64> A(na), B(nb), E(na), F(nb), T(nc);
    orc_prg_fill_buff(seed2, 0, 0, na, A.data(), nullptr);
    orc_prg_fill_buff(seed2, 0, na, nb, B.data(), nullptr);
    for (u64 i = 0; i < na; i++) E[i] = (i64)((u64)ef_peer[i] + (u64)ef_own[i]);
    for (u64 i = 0; i < nb; i++) F[i] = (i64)((u64)ef_peer[na + i] + (u64)ef_own[na + i]);
    for (u64 i = 0; i < nc; i++) T[i] = C[i];
    ring_gemm(E.data(), B.data(), T.data(), r, l, c, 0, 0, 0, 1);
    ring_gemm(A.data(), F.data(), T.data(), r, l, c, 0, 0, 0, 1);
    if (party) ring_gemm(E.data(), F.data(), T.data(), r, l, c, 0, 0, 0, 1);
    for (u64 i = 0; i < r; i++)
        for (u64 k = 0; k < c; k++) Z[layoutZ ? k * r + i : i * c + k] = T[i * c + k];
    return 0;
}
// vector_mul, party 2 (:226-252): A_j[k] = buf_j[k], B_j[k] = buf_j[k + n], C_0[k] = bufS[k]; (n_total, i0): a range
int orc_my_beaver_dealer_vecmul(u64 seed0, u64 seed1, u64 seedS, u64 n_total, u64 i0, u64 n, i64 *C0, i64 *C1) {
    std::vector<i64> A0(n), A1(n), B0(n), B1(n);
    orc_prg_fill_buff(seed0, 0, i0, n, A0.data(), nullptr);
    orc_prg_fill_buff(seed1, 0, i0, n, A1.data(), nullptr);
    orc_prg_fill_buff(seed0, 0, n_total + i0, n, B0.data(), nullptr);
    orc_prg_fill_buff(seed1, 0, n_total + i0, n, B1.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, i0, n, C0, nullptr);
    for (u64 k = 0; k < n; k++)
        C1[k] = (i64)(((u64)A0[k] + (u64)A1[k]) * ((u64)B0[k] + (u64)B1[k]) - (u64)C0[k]);
    return 0;
}
// party 0/1 before the round (:255-266): ef_out = (X - A)[n] | (Y - B)[n]
int orc_my_beaver_party_open_vecmul(const i64 *X, const i64 *Y, u64 seed2, u64 n_total, u64 i0, u64 n, i64 *ef_out) {
    std::vector<i64> A(n), B(n);
    orc_prg_fill_buff(seed2, 0, i0, n, A.data(), nullptr);
    orc_prg_fill_buff(seed2, 0, n_total + i0, n, B.data(), nullptr);
    for (u64 k = 0; k < n; k++) {
        ef_out[k] = (i64)((u64)X[k] - (u64)A[k]);
        ef_out[n + k] = (i64)((u64)Y[k] - (u64)B[k]);
    }
    return 0;
}
// party 0/1 after the round (:269-278)
int orc_my_beaver_party_finish_vecmul(int party, u64 seed2, const i64 *C, const i64 *ef_own, const i64 *ef_peer,
                                      u64 n_total, u64 i0, u64 n, i64 *Z) {
    std::vector<i64> A(n), B(n);
    orc_prg_fill_buff(seed2, 0, i0, n, A.data(), nullptr);
    orc_prg_fill_buff(seed2, 0, n_total + i0, n, B.data(), nullptr);
    for (u64 k = 0; k < n; k++) {
        const u64 E = (u64)ef_peer[k] + (u64)ef_own[k], F = (u64)ef_peer[n + k] + (u64)ef_own[n + k];
        Z[k] = (i64)((u64)C[k] + E * (u64)B[k] + (u64)A[k] * F + E * F * (u64)party);
    }
    return 0;
}
// privateCompare, party 0/1 (:293-357): s_k = (u16)buf01[64 i + k] % 67 (0 -> 1); beta = 0 compares against the
// bits of r, any other beta against the bits of r + 1 (no r = -1 branch: inputs are promised in [0, 2^64 - 2]).
int orc_my_pc_party(int party, const u16 *xbits, const i8 *beta, const i64 *r, u64 i0, u64 n, u64 seed01, u16 *d_out) {
    BuffReader rs(seed01, 64 * i0);
    const u64 party_id = (u64)party;
    for (u64 i = 0; i < n; i++) {
        u16 S[64], C[64], W[64], R[64], T[64];
        for (int k = 0; k < 64; k++) {
            S[k] = (u16)rs.next_int();
            S[k] = S[k] % P;
            if (0 == S[k]) S[k] = 1;
            R[k] = (u16)(((u64)r[i] >> k) & 1);
            T[k] = (u16)((((u64)r[i] + 1) >> k) & 1);
        }
        const u16 *X = xbits + 64 * i;
        u16 sum_w = 0;
        for (int k = 63; k != -1; k--) {
            if (beta[i] == 0) {
                C[k] = (u16)((X[k] * (P - 1)) % P + (u16)party_id * R[k] + (u16)party_id + sum_w);
                C[k] = C[k] % P;
                W[k] = (u16)(X[k] + (u16)party_id * R[k] + (P - 2) * R[k] * X[k]);
                W[k] = W[k] % P;
                sum_w = (u16)(sum_w + W[k]);
            } else {
                C[k] = (u16)((u64)(P - 1) * party_id * T[k] + X[k] + (u16)party_id + sum_w);
                C[k] = C[k] % P;
                W[k] = (u16)(X[k] + (u16)party_id * T[k] + (P - 2) * T[k] * X[k]);
                W[k] = W[k] % P;
                sum_w = (u16)(sum_w + W[k]);
            }
        }
        for (int k = 0; k < 64; k++) {
            u16 d = (u16)(S[k] * C[k]);
            d_out[64 * i + k] = d % P;
        }
    }
    return 0;
}
// privateCompare, party 2 (:402-426) is orc_snn_pc_helper.

// msb, party 2 before round 1 (:473-508): x = buf_0[k] % L_1 +L buf_1[k] % L_1; lsb shares as XOR bytes out of
// boolbufS[k]; Z_67 shares of the bits of x with share 0 = (u16)bufS[64 k + l] % 67.  Message to party j = xlsb_j | xbits_j.
int orc_my_msb_helper_deal(u64 seed0, u64 seed1, u64 seedS, u64 i0, u64 n, i8 *xlsb0, u16 *xbits0, i8 *xlsb1, u16 *xbits1) {
    std::vector<i64> w0(n), w1(n), dummy(n);
    orc_prg_fill_buff(seed0, 0, i0, n, w0.data(), nullptr);
    orc_prg_fill_buff(seed1, 0, i0, n, w1.data(), nullptr);
    orc_prg_fill_buff(seedS, 0, i0, n, dummy.data(), xlsb0);
    BuffReader rb(seedS, 64 * i0);
    for (u64 k = 0; k < n; k++) {
        const u64 x = add_L_1((u64)w0[k] % L_1, (u64)w1[k] % L_1);
        xlsb1[k] = (i8)((i8)(x & 1) ^ xlsb0[k]);
        for (int l = 0; l < 64; l++) {
            u16 s = (u16)rb.next_int();
            s = s % P;
            xbits0[64 * k + l] = s;
            xbits1[64 * k + l] = (u16)((u16)((x >> l) & 1) + P - s) % P;
        }
    }
    return 0;
}
// party 2 after round 2 (:512-525): XOR shares of beta' with share 0 = boolbufS[n + k]
int orc_my_msb_helper_beta(const i8 *beta1_plain, u64 n_total, u64 i0, u64 n, u64 seedS, i8 *b0, i8 *b1) {
    std::vector<i64> dummy(n);
    orc_prg_fill_buff(seedS, 0, n_total + i0, n, dummy.data(), b0);
    for (u64 k = 0; k < n; k++) b1[k] = (i8)(beta1_plain[k] ^ b0[k]);
    return 0;
}
// party 0/1 before round 1 (:446-452, :539-549): d = 2a, y = d +L d, x_j = buf2[k] % L_1, r_j = x_j +L y
int orc_my_msb_party_open(const i64 *A, u64 seed2, u64 i0, u64 n, i64 *r_out) {
    std::vector<i64> w(n);
    orc_prg_fill_buff(seed2, 0, i0, n, w.data(), nullptr);
    for (u64 k = 0; k < n; k++) {
        const u64 d = 2 * (u64)A[k];
        const u64 y = add_L_1(d, d);
        r_out[k] = (i64)add_L_1((u64)w[k] % L_1, y);
    }
    return 0;
}
// plaintext privateCompare inputs of round 2 (:542-544, :554-559): beta = boolbuf01[k], R = r_peer +L r_own
int orc_my_msb_party_pc_inputs(const i64 *r_own, const i64 *r_peer, u64 i0, u64 n, u64 seed01, i8 *beta_out, i64 *R_out) {
    std::vector<i64> dummy(n);
    orc_prg_fill_buff(seed01, 0, i0, n, dummy.data(), beta_out);
    for (u64 k = 0; k < n; k++) R_out[k] = (i64)add_L_1((u64)r_peer[k], (u64)r_own[k]);
    return 0;
}
// after round 3 (:573-581): XOR share of msb
int orc_my_msb_party_finish(int party, const i8 *beta1_j, const i8 *beta0, const i8 *xlsb_j, const i64 *R, u64 n, i8 *B) {
    for (u64 k = 0; k < n; k++) {
        const i8 b = party == 0 ? beta1_j[k] : (i8)(beta1_j[k] ^ beta0[k]);
        const i8 r0 = (i8)((u64)R[k] % 2);
        B[k] = party == 1 ? (i8)(b ^ xlsb_j[k]) : (i8)(b ^ xlsb_j[k] ^ r0);
    }
    return 0;
}
// bit_inject, party 2 (:703-750): b = boolbuf_0 ^ boolbuf_1, u = buf_0 + buf_1 (index k); shares of b*u, (1^b)*u,
// b_0*b_1 and b_0*(1^b_1) with share 0 = bufS[k], bufS[n+k], bufS[2n+k], bufS[3n+k].  Message to party j = V_j|W_j|S_j|T_j.
int orc_my_bit_inject_dealer(u64 seed0, u64 seed1, u64 seedS, u64 n_total, u64 i0, u64 n, i64 *V0, i64 *W0, i64 *S0,
                             i64 *T0, i64 *V1, i64 *W1, i64 *S1, i64 *T1) {
    std::vector<i64> U0(n), U1(n);
    std::vector<i8> B0(n), B1(n);
    orc_prg_fill_buff(seed0, 0, i0, n, U0.data(), B0.data());
    orc_prg_fill_buff(seed1, 0, i0, n, U1.data(), B1.data());
    orc_prg_fill_buff(seedS, 0, i0, n, V0, nullptr);
    orc_prg_fill_buff(seedS, 0, n_total + i0, n, W0, nullptr);
    orc_prg_fill_buff(seedS, 0, 2 * n_total + i0, n, S0, nullptr);
    orc_prg_fill_buff(seedS, 0, 3 * n_total + i0, n, T0, nullptr);
    for (u64 i = 0; i < n; i++) {
        const i8 b = (i8)(B0[i] ^ B1[i]);
        const u64 u = (u64)U0[i] + (u64)U1[i];
        V1[i] = (i64)((u64)(i64)b * u - (u64)V0[i]);
        W1[i] = (i64)((u64)(i64)(1 ^ b) * u - (u64)W0[i]);
        S1[i] = (i64)((u64)(i64)(B0[i] * B1[i]) - (u64)S0[i]);
        T1[i] = (i64)((u64)(i64)(B0[i] * (1 ^ B1[i])) - (u64)T0[i]);
    }
    return 0;
}
// party 0/1 before the round (:606-623): y_j = x_j - buf2[k], c_j = boolbuf2[k] ^ a_j; message to the peer = y_j | c_j
int orc_my_bit_inject_party_pre(const i64 *X, const i8 *A, u64 seed2, u64 i0, u64 n, i64 *y_out, i8 *c_out) {
    std::vector<i64> U(n);
    std::vector<i8> B(n);
    orc_prg_fill_buff(seed2, 0, i0, n, U.data(), B.data());
    for (u64 i = 0; i < n; i++) {
        y_out[i] = (i64)((u64)X[i] - (u64)U[i]);
        c_out[i] = (i8)(B[i] ^ A[i]);
    }
    return 0;
}
// party 0/1 after the round (:626-660)
int orc_my_bit_inject_party_post(int party, u64 seed2, u64 i0, u64 n, const i64 *y_own, const i8 *c_own, const i64 *y_peer,
                                 const i8 *c_peer, const i64 *V, const i64 *W, const i64 *S, const i64 *T, i64 *Z) {
    std::vector<i64> U(n);
    std::vector<i8> B(n);
    orc_prg_fill_buff(seed2, 0, i0, n, U.data(), B.data());
    for (u64 i = 0; i < n; i++) {
        const u64 y = (u64)y_peer[i] + (u64)y_own[i];
        const i8 cp = (i8)(c_peer[i] ^ c_own[i]);
        if (cp == 0)
            Z[i] = (i64)((u64)V[i] + y * ((u64)(i64)B[i] - 2 * (u64)S[i]));
        else
            Z[i] = (i64)((u64)W[i] + y * (((u64)party ^ (u64)(i64)B[i]) - 2 * (u64)T[i]));
    }
    return 0;
}
// reveal_vec_bools (:777-807): XOR of the two byte shares
int orc_xor_bools(const i8 *a, const i8 *b, i8 *out, u64 n) {
    for (u64 i = 0; i < n; i++) out[i] = (i8)(a[i] ^ b[i]);
    return 0;
}

} // extern "C"
