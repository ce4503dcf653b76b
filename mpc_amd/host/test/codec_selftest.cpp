// Wire codecs and containers on the host alone (no GPU): what the reference's test/test1.cpp:97-155 and :230-262
// check -- every codec gives its input back -- plus the byte positions the kernels rely on: matrix element (i, j) at
// byte (i * cols + j) * 8 whatever the storage order, Z_67 share k of element i at byte i * 128 + k * 2, one byte per
// boolean, no headers; sizes that are not multiples of the transposition tile; empty inputs.
#include <stdio.h>

#include "../utils/util.hpp"

static int failures = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);       \
            failures++;                                                 \
        }                                                               \
    } while (0)

static uint64_t next(uint64_t &s) { // any fixed sequence of distinct-looking words
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return s ^ (s >> 29);
}

static void matrices(size_t rows, size_t cols) {
    uint64_t s = rows * 1000 + cols;
    MatrixXl m(rows, cols);
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < cols; j++) m(i, j) = (int64_t)next(s);
    string wire;
    matrix2string(m, wire);
    CHECK(wire.size() == rows * cols * 8);
    const int64_t *w = (const int64_t *)wire.data();
    bool pos = true;
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < cols; j++) pos = pos && w[i * cols + j] == m(i, j);
    CHECK(pos);
    MatrixXl back(rows, cols);
    string2matrix(wire, back);
    CHECK(back == m);
    vector<int64_t> flat;
    matrix2vector(m, flat);
    CHECK(flat.size() == rows * cols && (flat.empty() || memcmp(flat.data(), wire.data(), wire.size()) == 0));
    MatrixXl back2(rows, cols);
    vector2matrix(flat, back2);
    CHECK(back2 == m);
    VecI64 arr;
    matrix2vecI64(m, arr);
    CHECK((size_t)arr.size() == rows * cols);
    bool same = true;
    for (size_t k = 0; k < rows * cols; k++) same = same && arr[k] == flat[k];
    CHECK(same);
    MatrixXl back3(rows, cols);
    vecI642matrix(arr, back3);
    CHECK(back3 == m);
}

static void vectors(size_t n) {
    uint64_t s = 77 + n;
    vector<int64_t> v(n);
    VecI64 a(n);
    vector<zpshare> z(n);
    vecbool b(n);
    for (size_t i = 0; i < n; i++) {
        v[i] = (int64_t)next(s);
        a[i] = (int64_t)next(s);
        b[i] = (int8_t)(next(s) & 1);
        for (size_t k = 0; k < 64; k++) z[i][k] = (uint16_t)(next(s) % P);
    }
    string wire;
    vector2string(v, wire);
    CHECK(wire.size() == 8 * n && (n == 0 || ((const int64_t *)wire.data())[n - 1] == v[n - 1]));
    vector<int64_t> v2(3, 5); // a destination that held something else
    string2vector(wire, v2);
    CHECK(v2 == v);

    vecI642string(a, wire);
    CHECK(wire.size() == 8 * n);
    VecI64 a2(1);
    string2vecI64(wire, a2);
    bool same = (size_t)a2.size() == n;
    for (size_t i = 0; same && i < n; i++) same = a2[i] == a[i];
    CHECK(same);

    vecZpshare2string(z, wire);
    CHECK(wire.size() == 128 * n);
    bool pos = true;
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < 64; k++) {
            uint16_t x;
            memcpy(&x, wire.data() + i * 128 + k * 2, 2);
            pos = pos && x == z[i][k];
        }
    CHECK(pos);
    vector<zpshare> z2;
    string2vecZpshare(wire, z2);
    CHECK(z2 == z);

    vecbool2string(b, wire);
    CHECK(wire.size() == n);
    vecbool b2(2, 9);
    string2vecbool(wire, b2);
    CHECK(b2 == b);
}

// The container subset keeps large released blocks for the next container of the same size on the thread, copies and
// moves keep their contents, and a matrix larger than one transposition tile band round-trips through the wire order.
static void storage() {
    const size_t rows = 600, cols = 700; // 3.2 MiB: above the block cache's threshold
    const int64_t *first;
    {
        MatrixXl a(rows, cols);
        first = a.data();
        for (size_t i = 0; i < rows * cols; i++) a.data()[i] = (int64_t)(i * 0x9E3779B97F4A7C15ull);
        MatrixXl b = a;
        CHECK(b == a && b.data() != a.data());
        string wire;
        matrix2string(a, wire);
        CHECK(((const int64_t *)wire.data())[5 * cols + 3] == a(5, 3));
        MatrixXl c(rows, cols);
        string2matrix(wire, c);
        CHECK(c == a);
        MatrixXl d(std::move(c));
        CHECK(d == a && c.size() == 0);
    }
    MatrixXl again(rows, cols); // one of the released blocks of this size comes back
    CHECK(again.data() != nullptr);
    (void)first;
    VecI64 v(1 << 18), w;
    for (size_t i = 0; i < (size_t)v.size(); i++) v[i] = (int64_t)i;
    string s;
    vecI642string(v, s);
    string2vecI64(s, w);
    CHECK(w == v);
}

int main() {
    storage();
    const size_t shapes[][2] = {{0, 0}, {1, 1}, {1, 7}, {7, 1}, {5, 4}, {32, 32}, {33, 31}, {64, 65}, {100, 3}, {3, 100}, {129, 70}};
    for (auto &sh : shapes) matrices(sh[0], sh[1]);
    const size_t sizes[] = {0, 1, 2, 10, 63, 64, 65, 1000};
    for (size_t n : sizes) vectors(n);
    CHECK(P == 67 && L_1 == 0xFFFFFFFFFFFFFFFFull && sizeof(zpshare) == 128);
    if (failures) {
        printf("codec_selftest: %d check(s) FAILED\n", failures);
        return 1;
    }
    printf("codec_selftest: all checks passed\n");
    return 0;
}
