// Self-test of the host engines: the three parties run as three threads of this process over the
// in-process messenger, every engine method is exercised the way the reference's own tests exercise it
// (test/test_ABY3.cpp, test/test_securenn.cpp: reveal(op(share(x))) == op(x)), and share checksums are
// printed as "KAT <name> <party> <hex>" lines so that a harness can compare them with known answers.
// Exit code 0 = every check passed.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>

#include "../mpc_engine/ABY3Engine.hpp"
#include "../mpc_engine/MyEngine.hpp"
#include "../mpc_engine/SecurennEngine.hpp"
#include "../utils/InProcMessenger.hpp"
#include "../utils/util.hpp"

static std::atomic<int> failures{0};
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            failures++;                                                              \
        }                                                                            \
    } while (0)

static uint64_t fnv(const int64_t *w, size_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n; i++) {
        h ^= (uint64_t)w[i];
        h *= 0x100000001B3ull;
    }
    return h;
}
static uint64_t fnv_rowmajor(const MatrixXl &m) {
    vector<int64_t> v;
    matrix2vector(m, v);
    return fnv(v.data(), v.size());
}

static void aby3_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    ABY3Engine e(&msgr, p);
    // share -> reveal identities (vector, matrix, booleans)
    {
        PRgenerator g(20, 0);
        const size_t n = 10;
        VecI64 X(n), Xr;
        for (size_t i = 0; i < n; i++) X[i] = g.pop_int64();
        array<VecI64, 2> Xj;
        e.share(X, Xj);
        e.reveal(Xj, Xr);
        CHECK(Xr == X);
        MatrixXl M(5, 4), Mr;
        for (size_t i = 0; i < 5; i++)
            for (size_t j = 0; j < 4; j++) M(i, j) = g.pop_int64();
        array<MatrixXl, 2> Mj;
        e.share_matrix(M, Mj);
        e.reveal_matrix(Mj, Mr);
        CHECK(Mr == M);
        vecbool B(n), Br;
        for (size_t i = 0; i < n; i++) B[i] = g.pop_bool();
        array<vecbool, 2> Bj;
        e.share_bools(B, Bj);
        e.reveal_bools(Bj, Br);
        CHECK(Br == B);
    }
    // products
    {
        PRgenerator g(31, 0);
        MatrixXl X(5, 4), Y(4, 3), Z;
        for (size_t i = 0; i < 5; i++)
            for (size_t j = 0; j < 4; j++) X(i, j) = g.pop_int64();
        for (size_t i = 0; i < 4; i++)
            for (size_t j = 0; j < 3; j++) Y(i, j) = g.pop_int64();
        array<MatrixXl, 2> Xj, Yj, Zj;
        e.share_matrix(X, Xj);
        e.share_matrix(Y, Yj);
        e.matrix_mul(Xj, Yj, Zj);
        e.reveal_matrix(Zj, Z);
        CHECK(Z == X * Y);
        const size_t n = 10;
        VecI64 U(n), V(n), Wv;
        for (size_t i = 0; i < n; i++) {
            U[i] = g.pop_int64();
            V[i] = g.pop_int64();
        }
        array<VecI64, 2> Uj, Vj, Wj;
        e.share(U, Uj);
        e.share(V, Vj);
        e.vector_mul(Uj, Vj, Wj);
        e.reveal(Wj, Wv);
        CHECK(Wv == U * V);
        vecbool A(100), B(100), C;
        for (size_t i = 0; i < 100; i++) {
            A[i] = g.pop_bool();
            B[i] = g.pop_bool();
        }
        array<vecbool, 2> Aj, Bj, Cj;
        e.share_bools(A, Aj);
        e.share_bools(B, Bj);
        e.vecbool_mul(Aj, Bj, Cj);
        e.reveal_bools(Cj, C);
        for (size_t i = 0; i < 100; i++) CHECK(C[i] == A[i] * B[i]);
    }
    // msb / bit_inject / trunc
    {
        PRgenerator g(21, 0);
        const size_t n = 1000;
        VecI64 X(n), Wr;
        for (size_t i = 0; i < n; i++) X[i] = g.pop_int64();
        X[0] = 0, X[1] = -1, X[2] = 1, X[3] = INT64_MIN, X[4] = INT64_MAX, X[5] = -2;
        array<VecI64, 2> Xj, Gj, Pj, Wj;
        array<vecbool, 2> Zj;
        vecbool Z;
        e.share(X, Xj);
        e.add_prepare(Xj, Gj, Pj);
        e.msb(Gj, Pj, Zj);
        e.reveal_bools(Zj, Z);
        for (size_t i = 0; i < n; i++) CHECK(Z[i] == (X[i] < 0));
        e.bit_inject(Xj, Zj, Wj);
        e.reveal(Wj, Wr);
        for (size_t i = 0; i < n; i++) CHECK(Wr[i] == X[i] * Z[i]);
        printf("KAT aby3_relu_1000_C0 %zu %016lX\n", p, (unsigned long)fnv(Wj[0].data(), n));
        VecI64 T(10), Tr;
        for (size_t i = 0; i < 10; i++) T[i] = g.pop_int64() >> 32;
        array<VecI64, 2> Tj, Sj;
        e.share(T, Tj);
        e.trunc(Tj, Sj, 16);
        e.reveal(Sj, Tr);
        for (size_t i = 0; i < 10; i++) CHECK(Tr[i] == (T[i] >> 16) || Tr[i] == (T[i] >> 16) - 1);
    }
}

// known-answer run on FRESH engines: X then Y 64x64 row-major from PRgenerator(20,0)
static void aby3_kat_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    ABY3Engine e(&msgr, p);
    PRgenerator g(20, 0);
    const size_t n = 64;
    MatrixXl X(n, n), Y(n, n), Z;
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) X(i, j) = g.pop_int64();
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) Y(i, j) = g.pop_int64();
    array<MatrixXl, 2> Xj, Yj, Zj;
    e.share_matrix(X, Xj);
    e.share_matrix(Y, Yj);
    e.matrix_mul(Xj, Yj, Zj);
    e.reveal_matrix(Zj, Z);
    CHECK(Z == X * Y);
    printf("KAT aby3_matmul_64_Z0 %zu %016lX\n", p, (unsigned long)fnv_rowmajor(Zj[0]));
    printf("KAT aby3_matmul_64_Zplain %zu %016lX\n", p, (unsigned long)fnv_rowmajor(Z));
}

static void snn_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    SecurennEngine e(&msgr, p);
    {
        // msb at n = 1000 on fresh engines: a = pop/2 from PRgenerator(22,0) after 67000 pops
        PRgenerator g(22, 0);
        const size_t n = 1000;
        for (size_t i = 0; i < 67 * n; i++) g.pop_int64();
        vector<int64_t> A(n), Aj, Bj, B;
        for (auto &a : A) a = g.pop_int64() / 2;
        e.share(A, Aj);
        e.msb(Aj, Bj);
        e.reveal(Bj, B);
        for (size_t i = 0; i < n; i++) CHECK(B[i] == (A[i] < 0));
        if (p != 2) printf("KAT snn_msb_1000_B %zu %016lX\n", p, (unsigned long)fnv(Bj.data(), n));
    }
    {
        vector<int64_t> X(10), Xj, Xr;
        for (size_t i = 0; i < 10; i++) X[i] = (int64_t)(i * 0x123456789ull) - 77;
        e.share(X, Xj);
        e.reveal(Xj, Xr);
        CHECK(Xr == X);
    }
    {
        PRgenerator g(33 + p, 0); // each party brings its own additive shares
        const size_t n = 5;
        vector<int64_t> Xj(n), Yj(n), Zj, X, Y, Z;
        for (size_t i = 0; i < n; i++) {
            Xj[i] = g.pop_int64();
            Yj[i] = g.pop_int64();
        }
        e.vector_mul(Xj, Yj, Zj);
        e.reveal(Xj, X);
        e.reveal(Yj, Y);
        e.reveal(Zj, Z);
        for (size_t i = 0; i < n; i++) CHECK((uint64_t)Z[i] == (uint64_t)X[i] * (uint64_t)Y[i]);
        const size_t rows = 3, lens = 4, cols = 5;
        MatrixXl Mx(rows, lens), My(lens, cols), Mz, Px, Py, Pz;
        for (size_t i = 0; i < rows; i++)
            for (size_t j = 0; j < lens; j++) Mx(i, j) = g.pop_int64();
        for (size_t i = 0; i < lens; i++)
            for (size_t j = 0; j < cols; j++) My(i, j) = g.pop_int64();
        e.matrix_mul(Mx, My, Mz);
        e.reveal_matrix(Mx, Px);
        e.reveal_matrix(My, Py);
        e.reveal_matrix(Mz, Pz);
        CHECK(Pz == Px * Py);
    }
    {
        // privateCompare with the r = -1 corner every fifth element
        PRgenerator g(22, 0);
        const size_t n = 50;
        vector<int64_t> X(n), R(n);
        vector<int8_t> B1(n), B2(n), Bp;
        vector<zpshare> xs[2];
        xs[0].resize(n);
        xs[1].resize(n);
        for (size_t i = 0; i < n; i++) {
            X[i] = g.pop_int64();
            R[i] = g.pop_int64();
            if (i % 5 == 2) R[i] = -1;
            B1[i] = g.pop_bool();
            for (size_t k = 0; k < 64; k++) {
                xs[0][i][k] = (uint16_t)g.pop_int64() % P;
                xs[1][i][k] = (uint16_t)((((uint64_t)X[i] >> k) & 1) + P - xs[0][i][k]) % P;
            }
        }
        vector<zpshare> back;
        e.reveal_zp(xs[p == 2 ? 0 : p], back);
        e.privateCompare(xs[p == 2 ? 0 : p], B1, R, B2);
        e.reveal_vec_bools(p == 2 ? B2 : B1, Bp);
        if (p != 2)
            for (size_t i = 0; i < n; i++) {
                CHECK(Bp[i] == ((uint64_t)X[i] > (uint64_t)R[i]));
                for (size_t k = 0; k < 64; k++) CHECK(back[i][k] == (((uint64_t)X[i] >> k) & 1));
            }
    }
    {
        // shareConvert + msb_L_1 on the corner values of test/test_securenn.cpp:247-251
        PRgenerator g(44, 0);
        const size_t n = 10;
        vector<int64_t> A(n), Aj, Yj, Y, Bj, B;
        for (auto &a : A) {
            a = g.pop_int64();
            if (a == -1) a = 1;
        }
        A[0] = 0, A[1] = INT64_MIN, A[2] = -2, A[3] = 1, A[4] = INT64_MAX;
        e.share(A, Aj);
        e.shareConvert(Aj, Yj);
        e.reveal_L_1(Yj, Y);
        if (p != 2) CHECK(Y == A);
        e.msb_L_1(Yj, Bj);
        e.reveal(Bj, B);
        for (size_t i = 0; i < n; i++) CHECK(B[i] == (Y[i] < 0));
    }
    {
        // the reference benchmark's own size, which needs 256000 buffer words (> BUFF_NUM)
        PRgenerator g(23, 0);
        const size_t n = 2000;
        vector<int64_t> A(n), Aj, Bj, B, Mj, M;
        for (auto &a : A) a = g.pop_int64() / 2;
        e.share(A, Aj);
        e.msb(Aj, Bj);
        e.vector_mul(Aj, Bj, Mj); // a * [a < 0], ReLU(a) = a - that
        e.reveal(Bj, B);
        e.reveal(Mj, M);
        for (size_t i = 0; i < n; i++) CHECK(B[i] == (A[i] < 0) && M[i] == A[i] * B[i]);
    }
}

// the checks of test/test_Myengine.cpp, plus the benchmark's size (n = 2000, beyond BUFF_NUM for privateCompare)
static void my_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    MyEngine e(&msgr, p);
    PRgenerator g(55, 0);
    {
        const size_t n = 10;
        VecI64 X(n), Y(n), Xj, Yj, Zj, Xr, Z;
        for (size_t i = 0; i < n; i++) {
            X[i] = g.pop_int64();
            Y[i] = g.pop_int64();
        }
        e.share(X, Xj);
        e.share(Y, Yj);
        e.reveal(Xj, Xr);
        CHECK(Xr == X);
        e.vector_mul(Xj, Yj, Zj);
        e.reveal(Zj, Z);
        CHECK(Z == X * Y);
        const size_t rows = 9, lens = 11, cols = 10;
        MatrixXl Mx(rows, lens), My(lens, cols), Mxj, Myj, Mzj, Mr, Mz;
        for (size_t i = 0; i < rows; i++)
            for (size_t k = 0; k < lens; k++) Mx(i, k) = g.pop_int64();
        for (size_t i = 0; i < lens; i++)
            for (size_t k = 0; k < cols; k++) My(i, k) = g.pop_int64();
        e.share_matrix(Mx, Mxj);
        e.share_matrix(My, Myj);
        e.reveal_matrix(Mxj, Mr);
        CHECK(Mr == Mx);
        e.matrix_mul(Mxj, Myj, Mzj);
        e.reveal_matrix(Mzj, Mz);
        CHECK(Mz == Mx * My);
    }
    {
        // privateCompare on the corner values of test/test_Myengine.cpp:101-109 (inputs in [0, 2^64 - 2])
        const size_t n = 10;
        VecI64 X(n), R(n);
        for (size_t i = 0; i < n; i++) {
            X[i] = (int64_t)((uint64_t)g.pop_int64() % L_1);
            R[i] = (int64_t)((uint64_t)g.pop_int64() % L_1);
        }
        X[0] = 0, X[1] = -2, R[3] = 0, R[4] = -2, X[5] = R[5];
        X[6] = (int64_t)(((uint64_t)R[6] + 1) % L_1);
        X[7] = (int64_t)(((uint64_t)R[7] - 1) % L_1);
        vector<zpshare> xs[2], plain(n);
        xs[0].resize(n);
        xs[1].resize(n);
        vecbool B1(n), B2(n);
        for (size_t i = 0; i < n; i++)
            for (size_t k = 0; k < 64; k++) {
                plain[i][k] = ((uint64_t)X[i] >> k) & 1;
                xs[0][i][k] = (uint16_t)g.pop_int64() % P;
                xs[1][i][k] = (uint16_t)(plain[i][k] + P - xs[0][i][k]) % P;
            }
        for (size_t i = 0; i < n; i++) B1[i] = g.pop_bool();
        e.privateCompare(p == 2 ? plain : xs[p], B1, R, B2);
        if (p == 2)
            for (size_t i = 0; i < n; i++) CHECK(((uint64_t)X[i] > (uint64_t)R[i]) == (B1[i] ^ B2[i]));
    }
    for (size_t n : {(size_t)10, (size_t)2000}) {
        VecI64 X(n), Xj, Zj, Z;
        vecbool Aj, A;
        for (size_t i = 0; i < n; i++) X[i] = g.pop_int64() / 2;
        X[0] = 0, X[1] = -1, X[2] = -2, X[3] = -((int64_t)1 << 62), X[4] = ((int64_t)1 << 62) - 1;
        e.share(X, Xj);
        if (p != 2) Xj[2] = -1; // test/test_Myengine.cpp:170: both shares -1, the value stays -2
        e.msb(Xj, Aj);
        e.reveal_vec_bools(Aj, A);
        for (size_t i = 0; i < n; i++) CHECK(A[i] == (X[i] < 0));
        e.bit_inject(Xj, Aj, Zj);
        e.reveal(Zj, Z);
        for (size_t i = 0; i < n; i++) CHECK(Z[i] == X[i] * A[i]);
        if (p != 2 && n == 2000) printf("KAT my_relu_2000_Z %zu %016lX\n", p, (unsigned long)fnv(Zj.data(), n));
    }
}

// relu as one protocol against the calls it is made of followed by the subtraction, each on FRESH engines: the same
// share words, the same generator positions afterwards; and matrix_mul with a pinned operand against the plain call
static void aby3_additions_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    PRgenerator g(21, 0);
    const size_t n = 3001; // not a multiple of the 64-element range granule
    VecI64 X(n), Rr;
    for (size_t i = 0; i < n; i++) X[i] = g.pop_int64();
    X[0] = 0, X[1] = -1, X[2] = 1, X[3] = INT64_MIN, X[4] = INT64_MAX;
    array<VecI64, 2> R1, R2;
    uint64_t pops1[3], pops2[3];
    {
        ABY3Engine e(&msgr, p);
        array<VecI64, 2> Xj, Gj, Pj, Cj;
        array<vecbool, 2> Zj;
        e.share(X, Xj);
        e.add_prepare(Xj, Gj, Pj);
        e.msb(Gj, Pj, Zj);
        e.bit_inject(Xj, Zj, Cj);
        for (int k = 0; k < 2; k++) {
            R1[k].resize(n);
            for (size_t i = 0; i < n; i++) R1[k][i] = (int64_t)((uint64_t)Xj[k][i] - (uint64_t)Cj[k][i]);
        }
        for (int k = 0; k < 3; k++) pops1[k] = e.prgs[k].pops;
    }
    {
        ABY3Engine e(&msgr, p);
        array<VecI64, 2> Xj;
        e.share(X, Xj);
        e.relu(Xj, R2);
        for (int k = 0; k < 3; k++) pops2[k] = e.prgs[k].pops;
        e.reveal(R2, Rr);
    }
    CHECK(R1[0] == R2[0] && R1[1] == R2[1]);
    for (int k = 0; k < 3; k++) CHECK(pops1[k] == pops2[k]);
    for (size_t i = 0; i < n; i++) CHECK(Rr[i] == (X[i] < 0 ? 0 : X[i]));
    printf("KAT aby3_relu_fused_3001_R0 %zu %016lX\n", p, (unsigned long)fnv(R2[0].data(), n));
    {
        // ragged and tiny vectors (fewer elements than one 64-element range, none at all)
        ABY3Engine e(&msgr, p);
        for (size_t m : {(size_t)0, (size_t)1, (size_t)63, (size_t)65, (size_t)130}) {
            VecI64 Y(m), Yr;
            for (size_t i = 0; i < m; i++) Y[i] = g.pop_int64();
            array<VecI64, 2> Yj, Rj;
            e.share(Y, Yj);
            e.relu(Yj, Rj);
            e.reveal(Rj, Yr);
            CHECK((size_t)Yr.size() == m);
            for (size_t i = 0; i < m; i++) CHECK(Yr[i] == (Y[i] < 0 ? 0 : Y[i]));
        }
    }
    {
        ABY3Engine e(&msgr, p);
        PRgenerator h(32, 0);
        MatrixXl A(37, 50), B(50, 29);
        for (size_t i = 0; i < 37; i++)
            for (size_t j = 0; j < 50; j++) A(i, j) = h.pop_int64();
        for (size_t i = 0; i < 50; i++)
            for (size_t j = 0; j < 29; j++) B(i, j) = h.pop_int64();
        array<MatrixXl, 2> Aj, Bj, Z1, Z2, Z3;
        e.share_matrix(A, Aj);
        e.share_matrix(B, Bj);
        e.pin(Bj);
        e.matrix_mul(Aj, Bj, Z1); // B from its device copy
        e.pin(Aj);
        e.matrix_mul(Aj, Bj, Z2); // both
        e.unpin(Aj);
        e.unpin(Bj);
        e.matrix_mul(Aj, Bj, Z3); // neither
        MatrixXl P1, P2, P3;
        e.reveal_matrix(Z1, P1);
        e.reveal_matrix(Z2, P2);
        e.reveal_matrix(Z3, P3);
        CHECK(P1 == A * B && P2 == P1 && P3 == P1);
    }
}
static void snn_additions_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    PRgenerator g(23, 0);
    const size_t n = 2001;
    vector<int64_t> A(n), R1(n), R2, Rr;
    for (auto &a : A) a = g.pop_int64() / 2;
    A[0] = 0, A[1] = -1, A[2] = 1;
    {
        SecurennEngine e(&msgr, p);
        vector<int64_t> Aj, Bj, Mj;
        e.share(A, Aj);
        e.msb(Aj, Bj);
        e.vector_mul(Aj, Bj, Mj);
        for (size_t i = 0; i < n; i++) R1[i] = p == 2 ? 0 : (int64_t)((uint64_t)Aj[i] - (uint64_t)Mj[i]);
    }
    {
        SecurennEngine e(&msgr, p);
        vector<int64_t> Aj;
        e.share(A, Aj);
        e.relu(Aj, R2);
        e.reveal(R2, Rr);
    }
    CHECK(R1 == R2);
    if (p != 2)
        for (size_t i = 0; i < n; i++) CHECK(Rr[i] == (A[i] < 0 ? 0 : A[i]));
    if (p != 2) printf("KAT snn_relu_fused_2001_R %zu %016lX\n", p, (unsigned long)fnv(R2.data(), n));
    {
        SecurennEngine e(&msgr, p);
        for (size_t m : {(size_t)0, (size_t)1, (size_t)63, (size_t)65, (size_t)130}) {
            vector<int64_t> Y(m), Yj, Rj, Yr;
            for (auto &y : Y) y = g.pop_int64() / 2;
            e.share(Y, Yj);
            e.relu(Yj, Rj);
            e.reveal(Rj, Yr);
            if (p != 2) {
                CHECK(Yr.size() == m);
                for (size_t i = 0; i < m; i++) CHECK(Yr[i] == (Y[i] < 0 ? 0 : Y[i]));
            }
        }
    }
    {
        SecurennEngine e(&msgr, p);
        PRgenerator h(34 + p, 0);
        MatrixXl Mx(13, 21), My(21, 8), Z1, Z2, P1, P2, Px, Py;
        for (size_t i = 0; i < 13; i++)
            for (size_t j = 0; j < 21; j++) Mx(i, j) = h.pop_int64();
        for (size_t i = 0; i < 21; i++)
            for (size_t j = 0; j < 8; j++) My(i, j) = h.pop_int64();
        e.pin(My);
        e.matrix_mul(Mx, My, Z1);
        e.unpin(My);
        e.matrix_mul(Mx, My, Z2);
        e.reveal_matrix(Mx, Px);
        e.reveal_matrix(My, Py);
        e.reveal_matrix(Z1, P1);
        e.reveal_matrix(Z2, P2);
        CHECK(P1 == Px * Py && P2 == P1);
    }
}

// every method on empty vectors and on matrices without rows or without columns: the reference's loops simply do not
// run, the results are empty, the rounds still take place (all three parties stay in step)
static void empty_inputs_party(InProcHub *hub, size_t p) {
    InProcMessenger msgr(hub, p);
    {
        ABY3Engine e(&msgr, p);
        VecI64 X, R;
        array<VecI64, 2> Xj, Gj, Pj, Cj, Tj, Vj;
        array<vecbool, 2> Zj, Bj;
        vecbool B, Bo;
        e.share(X, Xj);
        e.add_prepare(Xj, Gj, Pj);
        e.msb(Gj, Pj, Zj);
        e.bit_inject(Xj, Zj, Cj);
        e.trunc(Xj, Tj, 16);
        e.vector_mul(Xj, Xj, Vj);
        e.relu(Xj, Cj);
        e.reveal(Cj, R);
        e.share_bools(B, Bj);
        e.vecbool_mul(Bj, Bj, Zj);
        e.reveal_bools(Zj, Bo);
        e.reveal_bits(Xj, R);
        CHECK(R.size() == 0 && Bo.size() == 0 && Cj[0].size() == 0 && Zj[1].size() == 0);
        for (size_t shape = 0; shape < 3; shape++) {
            const size_t r = shape == 0 ? 0 : 3, l = shape == 1 ? 0 : 4, c = shape == 2 ? 0 : 5;
            MatrixXl A(r, l), Bm(l, c), P;
            for (size_t i = 0; i < r; i++)
                for (size_t j = 0; j < l; j++) A(i, j) = (int64_t)(i + 2 * j + 1);
            for (size_t i = 0; i < l; i++)
                for (size_t j = 0; j < c; j++) Bm(i, j) = (int64_t)(3 * i + j + 1);
            array<MatrixXl, 2> Aj, Bmj, Pj2;
            e.share_matrix(A, Aj);
            e.share_matrix(Bm, Bmj);
            e.matrix_mul(Aj, Bmj, Pj2);
            e.reveal_matrix(Pj2, P);
            CHECK((size_t)P.rows() == r && (size_t)P.cols() == c);
            for (size_t i = 0; i < r; i++)
                for (size_t j = 0; j < c; j++) CHECK(P(i, j) == 0); // l = 0: the empty sum
        }
    }
    {
        SecurennEngine e(&msgr, p);
        SecurennEngine::Words X, Xs, Y, R;
        SecurennEngine::Bits Bo;
        e.share(X, Xs);
        e.vector_mul(Xs, Xs, Y);
        e.msb(Xs, Y);
        e.relu(Xs, Y);
        e.shareConvert(Xs, Y);
        e.msb_L_1(Xs, Y);
        e.reveal(Y, R);
        vector<zpshare> xb;
        e.privateCompare(xb, Bo, X, Bo);
        CHECK(R.size() == 0 && Bo.size() == 0);
        MatrixXl A(0, 4), Bm(4, 5), Z, P;
        e.matrix_mul(A, Bm, Z);
        e.reveal_matrix(Z, P);
        CHECK(P.rows() == 0 && P.cols() == 5);
    }
    {
        MyEngine e(&msgr, p);
        VecI64 X, Xs, Z, R;
        vecbool A;
        e.share(X, Xs);
        e.msb(Xs, A);
        e.bit_inject(Xs, A, Z);
        e.vector_mul(Xs, Xs, Z);
        e.reveal(Z, R);
        CHECK(R.size() == 0 && A.size() == 0);
        MatrixXl M(0, 4), Bm(4, 5), Zm, P;
        e.matrix_mul(M, Bm, Zm);
        e.reveal_matrix(Zm, P);
        CHECK(P.rows() == 0 && P.cols() == 5);
    }
}

template <class F> static void run3(F party) {
    InProcHub hub;
    std::thread t0(party, &hub, 0), t1(party, &hub, 1), t2(party, &hub, 2);
    t0.join();
    t1.join();
    t2.join();
}

// containers: copy / move behaviour callers rely on (after `X = std::move(Z)` Eigen leaves the old X in Z), and the
// wire codecs' round trip through the row-major order
static void container_checks() {
    MatrixXl a(3, 2), b(5, 4);
    for (size_t i = 0; i < 3; i++)
        for (size_t j = 0; j < 2; j++) a(i, j) = (int64_t)(10 * i + j);
    MatrixXl a0 = a;
    a = std::move(b);
    CHECK(a.rows() == 5 && a.cols() == 4 && b.rows() == 3 && b.cols() == 2 && b == a0);
    MatrixXl c(std::move(b));
    CHECK(c == a0 && b.rows() == 0 && b.cols() == 0 && b.size() == 0);
    string wire;
    matrix2string(c, wire);
    CHECK(wire.size() == 48 && ((const int64_t *)wire.data())[1] == c(0, 1) && ((const int64_t *)wire.data())[2] == c(1, 0));
    MatrixXl d(3, 2);
    string2matrix(wire, d);
    CHECK(d == c);
    VecI64 v(4), w(2);
    for (size_t i = 0; i < 4; i++) v[i] = (int64_t)i;
    w = std::move(v);
    CHECK(w.size() == 4 && v.size() == 2 && w[3] == 3);
}

int main() {
    container_checks();
    run3(aby3_kat_party);
    run3(aby3_party);
    run3(snn_party);
    run3(my_party);
    run3(aby3_additions_party);
    run3(snn_additions_party);
    run3(empty_inputs_party);
    if (failures.load()) {
        printf("host_selftest: %d check(s) FAILED\n", failures.load());
        return 1;
    }
    printf("host_selftest: all checks passed\n");
    return 0;
}
