// Benchmark driver for the GPU-backed engines: the compositions the reference times in
// benchmark/benchmark1.cpp (ABY3 matrix_mul + trunc, ABY3 add_prepare + msb + bit_inject, SecureNN matrix_mul
// + local >> 16, SecureNN msb + vector_mul) plus the 784-128-128-10 SecureNN MLP at batch 128, at sizes inside
// and far beyond the reference's BUFF_NUM limit.  Differences from the reference driver, on purpose:
//   * wall-clock time (steady_clock) instead of clock(): CPU time hides the time a party spends blocked;
//   * every result is revealed and checked against the plaintext computation;
//   * without arguments the three parties run as three threads of this process over InProcMessenger (all on
//     MPCX_DEVICE, default GPU 0); with a party id it is one party of a three-process run over SocketMessenger,
//     as the reference driver is.
// usage: benchmark1 [party_id] [--width W] [--vec N] [--repeat R] [--engine aby3|securenn|my] [--no-mlp | --mlp-only] [--pin]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>

#include "../mpc_engine/ABY3Engine.hpp"
#include "../mpc_engine/MyEngine.hpp"
#include "../mpc_engine/SecurennEngine.hpp"
#include "../utils/InProcMessenger.hpp"
#include "../utils/SocketMessenger.hpp"
#include "../utils/util.hpp"

struct Options {
    size_t width = 200, vec = 0, repeat = 10;
    bool mlp = true, mlp_only = false;
    bool pin = false; // keep the second operand of the matrix products (the "weights") on the device
    const char *engine = nullptr; // run only this engine's lines: "aby3", "securenn" or "my"
};
static std::atomic<int> failures{0};
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            if (failures++ < 10) fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
        }                                                                             \
    } while (0)

using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

static void report(size_t party, const char *what, double ms, double units, const char *unit) {
    if (party != 0) return;
    printf("[LOG]:\t%s\n", what);
    print_plus();
    printf("spent time: %.3f ms per repetition   (%.3g %s/s)\n", ms, units / (ms * 1e-3), unit);
    print_minus();
    printf("RESULT\t%s\tms=%.4f\trate=%.6g\tunit=%s/s\n", what, ms, units / (ms * 1e-3), unit); // one line for scripts
    fflush(stdout);
}

static void bench_aby3(VirtuleMessenger *messenger, size_t party, const Options &opt) {
    ABY3Engine engine(messenger, party);
    PRgenerator rnd(0x5EED, 0); // all parties draw the same plaintext
    const size_t w = opt.width, vec_size = opt.vec ? opt.vec : w * 10;
    double spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) { // repetition 0 warms up (checkpoint tables, device pool)
        MatrixXl X(w, w), Y(w, w), Zp;
        for (size_t i = 0; i < w; i++)
            for (size_t j = 0; j < w; j++) {
                X(i, j) = rnd.pop_int64() >> 48; // Q16 operands in [-0.5, 0.5): share-local truncation errs with probability ~|z| / 2^64
                Y(i, j) = rnd.pop_int64() >> 48;
            }
        array<MatrixXl, 2> Xj, Yj, Zj;
        engine.share_matrix(X, Xj);
        engine.share_matrix(Y, Yj);
        if (opt.pin) engine.pin(Yj); // the second operand as a resident weight matrix
        // the clock runs around the two engine calls only, as in the reference (benchmark/benchmark1.cpp:33-50)
        auto t0 = Clock::now();
        engine.matrix_mul(Xj, Yj, Zj);
        double ms = ms_since(t0);
        if (opt.pin) engine.unpin(Yj);
        array<VecI64, 2> Zv, Tv;
        for (size_t i = 0; i < 2; i++) matrix2vecI64(Zj[i], Zv[i]);
        t0 = Clock::now();
        engine.trunc(Zv, Tv, 16);
        ms += ms_since(t0);
        if (t) spent += ms;
        VecI64 T;
        engine.reveal(Tv, T);
        if (t == opt.repeat) { // check the last repetition: trunc may be one below the exact shift
            size_t bad = 0;
            if (w <= 1024) {
                MatrixXl Z = X * Y;
                for (size_t i = 0; i < w; i++)
                    for (size_t j = 0; j < w; j++) {
                        const int64_t want = Z(i, j) >> 16, got = T[i * w + j];
                        bad += !(got == want || got == want - 1);
                    }
            } else { // the plain product of wide matrices would take a host core a minute: 4096 sampled entries
                PRgenerator pick(0xC4EC, 0);
                for (size_t s = 0; s < 4096; s++) {
                    const size_t i = (uint64_t)pick.pop_int64() % w, j = (uint64_t)pick.pop_int64() % w;
                    uint64_t dot = 0;
                    for (size_t k = 0; k < w; k++) dot += (uint64_t)X(i, k) * (uint64_t)Y(k, j);
                    const int64_t want = (int64_t)dot >> 16, got = T[i * w + j];
                    bad += !(got == want || got == want - 1);
                }
            }
            CHECK(bad == 0);
        }
    }
    report(party, "bench_aby3 matrix_mul & trunction", spent / opt.repeat, 3.0 * 2 * w * w * w, "ring op");

    spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        VecI64 X(vec_size), C;
        for (size_t i = 0; i < vec_size; i++) X[i] = rnd.pop_int64();
        array<VecI64, 2> Xj, Gj, Pj, Cj;
        array<vecbool, 2> Zj;
        engine.share(X, Xj);
        auto t0 = Clock::now();
        engine.add_prepare(Xj, Gj, Pj);
        engine.msb(Gj, Pj, Zj);
        engine.bit_inject(Xj, Zj, Cj);
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            engine.reveal(Cj, C);
            size_t bad = 0;
            for (size_t i = 0; i < vec_size; i++) bad += C[i] != (X[i] < 0 ? X[i] : 0);
            CHECK(bad == 0);
        }
    }
    report(party, "bench_aby3 msb & bit_inject", spent / opt.repeat, (double)vec_size, "elem");

    // the same chain plus the subtraction as the engine's one relu protocol (an addition to the reference's surface)
    spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        VecI64 X(vec_size), R;
        for (size_t i = 0; i < vec_size; i++) X[i] = rnd.pop_int64();
        array<VecI64, 2> Xj, Rj;
        engine.share(X, Xj);
        auto t0 = Clock::now();
        engine.relu(Xj, Rj);
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            engine.reveal(Rj, R);
            size_t bad = 0;
            for (size_t i = 0; i < vec_size; i++) bad += R[i] != (X[i] < 0 ? 0 : X[i]);
            CHECK(bad == 0);
        }
    }
    report(party, "bench_aby3 relu", spent / opt.repeat, (double)vec_size, "elem");
}

static void bench_securenn(VirtuleMessenger *messenger, size_t party, const Options &opt) {
    SecurennEngine engine(messenger, party);
    PRgenerator rnd(0x5EED + 1, 0);
    const size_t w = opt.width, vec_size = opt.vec ? opt.vec : w * 10;
    double spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        vector<int64_t> Xp(w * w), Yp(w * w), Xv, Yv;
        for (size_t i = 0; i < w * w; i++) {
            Xp[i] = rnd.pop_int64() >> 48;
            Yp[i] = rnd.pop_int64() >> 48;
        }
        engine.share(Xp, Xv);
        engine.share(Yp, Yv);
        MatrixXl Xj(w, w), Yj(w, w), Zj;
        vector2matrix(Xv, Xj);
        vector2matrix(Yv, Yj);
        if (opt.pin) engine.pin(Yj);
        auto t0 = Clock::now();
        engine.matrix_mul(Xj, Yj, Zj);
        for (size_t i = 0; i < w; i++)
            for (size_t j = 0; j < w; j++) Zj(i, j) = Zj(i, j) >> 16; // benchmark/benchmark1.cpp:106-112
        const double ms = ms_since(t0);
        if (opt.pin) engine.unpin(Yj);
        if (t) spent += ms;
        if (t == opt.repeat) {
            MatrixXl Z, X(w, w), Y(w, w);
            engine.reveal_matrix(Zj, Z);
            vector2matrix(Xp, X);
            vector2matrix(Yp, Y);
            MatrixXl W = X * Y;
            size_t bad = 0;
            if (party != 2)
                for (size_t i = 0; i < w; i++)
                    for (size_t j = 0; j < w; j++) {
                        const int64_t d = Z(i, j) - (W(i, j) >> 16);
                        bad += !(d >= -1 && d <= 1); // local truncation of two additive shares
                    }
            CHECK(bad == 0);
        }
    }
    report(party, "bench_securenn matrix_mul & trunction", spent / opt.repeat, 3.0 * 2 * w * w * w, "ring op");

    spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        vector<int64_t> A(vec_size), Aj, Bj, Cj, C;
        for (auto &a : A) a = rnd.pop_int64() / 2;
        engine.share(A, Aj);
        auto t0 = Clock::now();
        engine.msb(Aj, Bj);
        engine.vector_mul(Aj, Bj, Cj);
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            engine.reveal(Cj, C);
            size_t bad = 0;
            if (party != 2)
                for (size_t i = 0; i < vec_size; i++) bad += C[i] != (A[i] < 0 ? A[i] : 0);
            CHECK(bad == 0);
        }
    }
    report(party, "bench_securenn msb & vector_mul", spent / opt.repeat, (double)vec_size, "elem");

    spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        vector<int64_t> A(vec_size), Aj, Rj, R;
        for (auto &a : A) a = rnd.pop_int64() / 2;
        engine.share(A, Aj);
        auto t0 = Clock::now();
        engine.relu(Aj, Rj);
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            engine.reveal(Rj, R);
            size_t bad = 0;
            if (party != 2)
                for (size_t i = 0; i < vec_size; i++) bad += R[i] != (A[i] < 0 ? 0 : A[i]);
            CHECK(bad == 0);
        }
    }
    report(party, "bench_securenn relu", spent / opt.repeat, (double)vec_size, "elem");
}

static void bench_me(VirtuleMessenger *messenger, size_t party, const Options &opt) {
    MyEngine engine(messenger, party);
    PRgenerator rnd(0x5EED + 2, 0);
    const size_t w = opt.width, vec_size = opt.vec ? opt.vec : w * 10;
    double spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        MatrixXl X(w, w), Y(w, w), Xj, Yj, Zj;
        for (size_t i = 0; i < w; i++)
            for (size_t k = 0; k < w; k++) {
                X(i, k) = rnd.pop_int64() >> 48;
                Y(i, k) = rnd.pop_int64() >> 48;
            }
        engine.share_matrix(X, Xj);
        engine.share_matrix(Y, Yj);
        auto t0 = Clock::now();
        engine.matrix_mul(Xj, Yj, Zj);
        for (size_t i = 0; i < w; i++)
            for (size_t j = 0; j < w; j++) Zj(i, j) = Zj(i, j) >> 16; // benchmark/benchmark1.cpp:168-174
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            MatrixXl Z, Wp = X * Y;
            engine.reveal_matrix(Zj, Z);
            size_t bad = 0;
            if (party != 2)
                for (size_t i = 0; i < w; i++)
                    for (size_t j = 0; j < w; j++) {
                        const int64_t d = Z(i, j) - (Wp(i, j) >> 16);
                        bad += !(d >= -1 && d <= 1);
                    }
            CHECK(bad == 0);
        }
    }
    report(party, "bench_me matrix_mul & trunction", spent / opt.repeat, 3.0 * 2 * w * w * w, "ring op");

    spent = 0;
    for (size_t t = 0; t <= opt.repeat; t++) {
        VecI64 X(vec_size), Xj, Zj, Z;
        vecbool Aj;
        for (size_t i = 0; i < vec_size; i++) X[i] = rnd.pop_int64() / 2;
        engine.share(X, Xj);
        auto t0 = Clock::now();
        engine.msb(Xj, Aj);
        engine.bit_inject(Xj, Aj, Zj);
        const double ms = ms_since(t0);
        if (t) spent += ms;
        if (t == opt.repeat) {
            engine.reveal(Zj, Z);
            size_t bad = 0;
            for (size_t i = 0; i < vec_size; i++) bad += Z[i] != (X[i] < 0 ? X[i] : 0);
            CHECK(bad == 0);
        }
    }
    report(party, "bench_me msb & bit_inject", spent / opt.repeat, (double)vec_size, "elem");
}

// One dense layer on additive shares: Z = X * W, rescaled by a local >> 16 on each share, then (optionally)
// ReLU(a) = a - a * msb(a), as the reference benchmark composes it (benchmark/benchmark1.cpp:104-136).
static void snn_dense(SecurennEngine &engine, const MatrixXl &Xj, const MatrixXl &Wj, bool relu, MatrixXl &out) {
    MatrixXl Zj;
    engine.matrix_mul(Xj, Wj, Zj);
    const size_t r = Zj.rows(), c = Zj.cols();
    vector<int64_t> a(r * c);
    for (size_t i = 0; i < r; i++)
        for (size_t j = 0; j < c; j++) a[i * c + j] = Zj(i, j) >> 16;
    if (relu) {
        vector<int64_t> r;
        engine.relu(a, r); // = msb, vector_mul(a, msb), a - that (benchmark/benchmark1.cpp:135-136), as one protocol
        a.swap(r);
    }
    out.resize(r, c);
    vector2matrix(a, out);
}

// 784-128-128-10 at batch 128; input and weights are PRgenerator(24,0).pop_int64() >> 48 in the order
// input, W1, W2, W3 (Q16 values in [-0.5, 0.5)).
static void bench_mlp(VirtuleMessenger *messenger, size_t party, const Options &opt) {
    SecurennEngine engine(messenger, party);
    const size_t batch = 128, dims[4] = {784, 128, 128, 10};
    PRgenerator rnd(24, 0);
    auto draw = [&](size_t r, size_t c, MatrixXl &M) {
        M.resize(r, c);
        for (size_t i = 0; i < r; i++)
            for (size_t j = 0; j < c; j++) M(i, j) = rnd.pop_int64() >> 48;
    };
    MatrixXl X, W[3], Xj, Wj[3];
    draw(batch, dims[0], X);
    for (int l = 0; l < 3; l++) draw(dims[l], dims[l + 1], W[l]);
    auto share_matrix = [&](const MatrixXl &M, MatrixXl &Mj) {
        vector<int64_t> v, vj;
        matrix2vector(M, v);
        engine.share(v, vj);
        Mj.resize(M.rows(), M.cols());
        vector2matrix(vj, Mj);
    };
    share_matrix(X, Xj);
    for (int l = 0; l < 3; l++) share_matrix(W[l], Wj[l]);
    for (int l = 0; l < 3; l++) engine.pin(Wj[l]); // the weights stay on the device for every pass
    double total = 0, layer_ms[3] = {0, 0, 0};
    MatrixXl act;
    const size_t reps = opt.repeat < 3 ? opt.repeat : 3;
    for (size_t t = 0; t <= reps; t++) {
        act = Xj;
        for (int l = 0; l < 3; l++) {
            auto t0 = Clock::now();
            MatrixXl next;
            snn_dense(engine, act, Wj[l], l < 2, next);
            act = next;
            const double ms = ms_since(t0);
            if (t) layer_ms[l] += ms, total += ms;
        }
    }
    for (int l = 0; l < 3; l++) engine.unpin(Wj[l]);
    MatrixXl out;
    engine.reveal_matrix(act, out);
    if (party != 2) {
        MatrixXl h = X;
        for (int l = 0; l < 3; l++) {
            MatrixXl z = h * W[l];
            for (size_t i = 0; i < (size_t)z.rows(); i++)
                for (size_t j = 0; j < (size_t)z.cols(); j++) {
                    int64_t v = z(i, j) >> 16;
                    z(i, j) = (l < 2 && v < 0) ? 0 : v;
                }
            h = z;
        }
        size_t bad = 0;
        for (size_t i = 0; i < batch; i++)
            for (size_t j = 0; j < dims[3]; j++) {
                const int64_t d = out(i, j) - h(i, j);
                bad += !(d >= -64 && d <= 64); // share-local truncation is off by at most one per layer
            }
        CHECK(bad == 0);
    }
    if (party == 0) {
        printf("[LOG]:\tbench_securenn MLP 784-128-128-10, batch 128 (matrix_mul, >>16, msb, vector_mul per layer)\n");
        print_plus();
        printf("latency: %.3f ms   layers: %.3f / %.3f / %.3f ms\n", total / reps, layer_ms[0] / reps, layer_ms[1] / reps,
               layer_ms[2] / reps);
        print_minus();
        printf("RESULT\tbench_securenn MLP\tms=%.4f\tlayers_ms=%.4f,%.4f,%.4f\n", total / reps, layer_ms[0] / reps,
               layer_ms[1] / reps, layer_ms[2] / reps);
        fflush(stdout);
    }
}

static void party_main(VirtuleMessenger *messenger, size_t party, const Options &opt) {
    const auto wanted = [&](const char *name) { return !opt.engine || !strcmp(opt.engine, name); };
    if (!opt.mlp_only) {
        if (wanted("aby3")) bench_aby3(messenger, party, opt);
        if (wanted("securenn")) bench_securenn(messenger, party, opt);
        if (wanted("my")) bench_me(messenger, party, opt);
    }
    if (opt.mlp && wanted("securenn")) bench_mlp(messenger, party, opt);
}

int main(int argc, char const *argv[]) {
    Options opt;
    long party = -1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--width") && i + 1 < argc)
            opt.width = strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--vec") && i + 1 < argc)
            opt.vec = strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc)
            opt.repeat = strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--engine") && i + 1 < argc)
            opt.engine = argv[++i];
        else if (!strcmp(argv[i], "--no-mlp"))
            opt.mlp = false;
        else if (!strcmp(argv[i], "--mlp-only"))
            opt.mlp_only = true;
        else if (!strcmp(argv[i], "--pin"))
            opt.pin = true;
        else if (argv[i][0] >= '0' && argv[i][0] <= '2' && argv[i][1] == 0)
            party = argv[i][0] - '0';
        else {
            printf("usage:\nbenchmark1 [party_id in {0,1,2}] [--width W] [--vec N] [--repeat R] [--engine aby3|securenn|my] [--no-mlp | --mlp-only] [--pin] [--pin]\n");
            return 1;
        }
    }
    if (opt.width == 0 || opt.repeat == 0) {
        printf("width and repeat must be positive\n");
        return 1;
    }
    if (party >= 0) {
        SocketMessenger socketMessenger((size_t)party);
        party_main(&socketMessenger, (size_t)party, opt);
    } else {
        InProcHub hub;
        std::thread th[3];
        for (size_t p = 0; p < 3; p++)
            th[p] = std::thread([&hub, p, &opt]() {
                InProcMessenger messenger(&hub, p);
                party_main(&messenger, p, opt);
            });
        for (auto &t : th) t.join();
    }
    if (failures.load()) {
        printf("benchmark1: %d result check(s) FAILED\n", failures.load());
        return 1;
    }
    printf("benchmark1: all results verified\n");
    return 0;
}
