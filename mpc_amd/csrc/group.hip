// Several GPUs behind one party: a group is one context per device plus one host thread per member.  The path
// shards without any exchange between the members -- matrix products by row blocks of X (Y replicated, the mask
// words of a block start row0 * c words into both pop streams), element-wise segments by contiguous element ranges
// (n_total, i0, n) -- and every member moves its own slice between host memory and its device, so the blocks of a
// message land in the messenger-facing host buffer directly, with no gather between the devices.
#include <cstdarg>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "mpcx_internal.h"

struct mpcx_group {
    std::vector<mpcx_ctx *> ctx;
    std::string err;
    // one worker per member, bound to the member's device; a call hands every worker one job and waits for all
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = false, quit = false;
        int rc = 0;
    };
    std::vector<std::unique_ptr<Worker>> workers;
};

// A replicated operand of matrix products (both shares of Y, row-major) that every member of a group holds on its
// device from mpcx_group_operand_upload until mpcx_group_operand_free, together with its residue planes where products
// of its shape run in the residue form (mpcx_aby3_operand_prepare).
struct mpcx_group_operand {
    u64 rows, cols;
    std::vector<void *> s0, s1; // per member
    std::vector<mpcx_aby3_operand *> prepared;
};

namespace {

void worker_main(mpcx_group::Worker *w, int device) {
    (void)hipSetDevice(device);
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const int rc = job();
        lk.lock();
        w->rc = rc;
        w->done = true;
        w->cv.notify_all();
    }
}

int group_fail(mpcx_group *g, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g)
        g->err = buf;
    else
        mpcx_fail(nullptr, "%s", buf);
    return 1;
}

// runs job(member) on every member's thread (member 0 on the caller's own thread) and collects the first error
template <class F> int group_run(mpcx_group *g, int members, F job) {
    for (int k = 1; k < members; k++) {
        mpcx_group::Worker &w = *g->workers[k];
        std::lock_guard<std::mutex> lk(w.mu);
        w.job = [=] { return job(k); };
        w.has_job = true;
        w.done = false;
        w.cv.notify_all();
    }
    int rc = 0;
    if (members > 0) {
        (void)hipSetDevice(g->ctx[0]->device);
        rc = job(0);
        if (rc) g->err = "member 0: " + g->ctx[0]->err;
    }
    for (int k = 1; k < members; k++) {
        mpcx_group::Worker &w = *g->workers[k];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        if (w.rc && !rc) {
            rc = w.rc;
            g->err = "member " + std::to_string(k) + ": " + g->ctx[k]->err;
        }
    }
    return rc;
}

} // namespace

extern "C" {

int mpcx_plan_split(uint64_t n, int parts, int part, uint64_t granule, uint64_t *first, uint64_t *count) {
    if (!first || !count || parts < 1 || part < 0 || part >= parts) return 1;
    if (granule == 0) granule = 1;
    const u64 units = ceil_div(n, granule);
    const u64 base = units / (u64)parts, rem = units % (u64)parts; // the first `rem` parts take one unit more
    const u64 u0 = (u64)part * base + ((u64)part < rem ? (u64)part : rem);
    const u64 un = base + ((u64)part < rem ? 1 : 0);
    u64 f = u0 * granule, e = (u0 + un) * granule;
    if (f > n) f = n;
    if (e > n) e = n;
    *first = f;
    *count = e - f;
    return 0;
}

int mpcx_group_init(const int *devices, int n, mpcx_group **out) {
    if (!out || !devices || n < 1 || n > 64) return group_fail(nullptr, "mpcx_group_init: bad arguments (n = %d)", n);
    mpcx_group *g = new mpcx_group();
    for (int k = 0; k < n; k++) {
        mpcx_ctx *c = nullptr;
        if (mpcx_init(devices[k], &c)) { // the text is in the init error slot
            for (mpcx_ctx *p : g->ctx) mpcx_shutdown(p);
            delete g;
            return 1;
        }
        g->ctx.push_back(c);
    }
    g->workers.resize(n);
    for (int k = 1; k < n; k++) {
        g->workers[k].reset(new mpcx_group::Worker());
        g->workers[k]->th = std::thread(worker_main, g->workers[k].get(), devices[k]);
    }
    *out = g;
    return 0;
}

int mpcx_group_shutdown(mpcx_group *g) {
    if (!g) return 0;
    for (size_t k = 1; k < g->workers.size(); k++) {
        mpcx_group::Worker &w = *g->workers[k];
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.quit = true;
            w.cv.notify_all();
        }
        w.th.join();
    }
    for (mpcx_ctx *c : g->ctx) mpcx_shutdown(c);
    delete g;
    return 0;
}

int mpcx_group_size(const mpcx_group *g) { return g ? (int)g->ctx.size() : 0; }
mpcx_ctx *mpcx_group_ctx(mpcx_group *g, int member) {
    return (g && member >= 0 && member < (int)g->ctx.size()) ? g->ctx[member] : nullptr;
}
const char *mpcx_group_last_error(mpcx_group *g) { return g ? g->err.c_str() : mpcx_last_error(nullptr); }

int mpcx_group_for_each(mpcx_group *g, uint64_t n, uint64_t granule, mpcx_member_fn fn, void *user) {
    if (!g || !fn) return group_fail(g, "mpcx_group_for_each: NULL argument");
    const int size = (int)g->ctx.size();
    // a vector too short to give every member a granule runs on the first members only
    int members = size;
    const u64 units = ceil_div(n, granule ? granule : 1);
    if (units < (u64)members) members = units ? (int)units : 1;
    return group_run(g, members, [=](int k) {
        u64 first = 0, count = 0;
        mpcx_plan_split(n, members, k, granule, &first, &count);
        return fn(g->ctx[k], k, first, count, user);
    });
}

// the common part of the two forms below: op == NULL uploads Y0 / Y1 (host) per call, otherwise the members' resident
// copies are used
static int group_aby3_matmul(mpcx_group *g, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                             const mpcx_group_operand *op, uint64_t r, uint64_t l, uint64_t c, int layout,
                             uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0) {
    int members = (int)g->ctx.size();
    const u64 granule = 128; // rows of a block tile of the digit-pair kernel
    if (ceil_div(r, granule) < (u64)members) members = (int)ceil_div(r, granule);
    return group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        u64 row0 = 0, rows = 0;
        mpcx_plan_split(r, members, k, granule, &row0, &rows);
        if (rows == 0) return 0;
        // device copies: the member's rows of X0 / X1 (row-major), all of Y0 / Y1 (row-major), its rows of Z0
        void *px0 = nullptr, *px1 = nullptr, *py0 = nullptr, *py1 = nullptr, *pz = nullptr, *tmp = nullptr;
        auto release = [&](int rc) {
            for (void *p : {px0, px1, py0, py1, pz, tmp}) (void)mpcx_free(ctx, p);
            return rc;
        };
#define G_TRY(call)                                                                                                    \
    do {                                                                                                               \
        if ((call) != 0) return release(1);                                                                            \
    } while (0)
        G_TRY(mpcx_malloc(ctx, 8 * rows * l, &px0));
        G_TRY(mpcx_malloc(ctx, 8 * rows * l, &px1));
        if (!op) {
            G_TRY(mpcx_malloc(ctx, 8 * l * c, &py0));
            G_TRY(mpcx_malloc(ctx, 8 * l * c, &py1));
        }
        G_TRY(mpcx_malloc(ctx, 8 * rows * c, &pz));
        if (layout == 0) {
            G_TRY(mpcx_h2d(ctx, px0, X0 + row0 * l, 8 * rows * l));
            G_TRY(mpcx_h2d(ctx, px1, X1 + row0 * l, 8 * rows * l));
            if (!op) {
                G_TRY(mpcx_h2d(ctx, py0, Y0, 8 * l * c));
                G_TRY(mpcx_h2d(ctx, py1, Y1, 8 * l * c));
            }
        } else {
            // column-major operands (how Eigen stores MatrixXl): the member's rows of X are a strided slab on the host --
            // `rows` consecutive words in each of the l columns -- which the host threads gather into the pinned staging
            // buffers as a rows x l column-major block; it is turned on the device.  Only the member's share of X
            // crosses the link.  (A 2-D hipMemcpy from pageable memory is not used: several host threads issuing it
            // beside staged copies of other contexts did not return.)
            const u64 big = (op || rows * l > l * c) ? rows * l : l * c;
            G_TRY(mpcx_malloc(ctx, 8 * big, &tmp));
            for (int s = 0; s < 2; s++) {
                const i64 *src = (s ? X1 : X0) + row0;
                G_TRY(mpcx_h2d_gather(ctx, tmp, src, 8 * r, 8 * rows, l));
                G_TRY(mpcx_relayout(ctx, (const i64 *)tmp, (i64 *)(s ? px1 : px0), rows, l, 1, 0));
            }
            if (!op) {
                G_TRY(mpcx_h2d(ctx, tmp, Y0, 8 * l * c));
                G_TRY(mpcx_relayout(ctx, (const i64 *)tmp, (i64 *)py0, l, c, 1, 0));
                G_TRY(mpcx_h2d(ctx, tmp, Y1, 8 * l * c));
                G_TRY(mpcx_relayout(ctx, (const i64 *)tmp, (i64 *)py1, l, c, 1, 0));
            }
        }
        if (op)
            G_TRY(mpcx_aby3_matmul_local_prepared(ctx, (const i64 *)px0, (const i64 *)px1, 0, op->prepared[k], rows, row0,
                                                  seed_next, pop_next, seed_prev, pop_prev, (i64 *)pz));
        else
            G_TRY(mpcx_aby3_matmul_local_rows(ctx, (const i64 *)px0, (const i64 *)px1, (const i64 *)py0, (const i64 *)py1,
                                              rows, l, c, row0, seed_next, pop_next, seed_prev, pop_prev, (i64 *)pz));
        G_TRY(mpcx_d2h(ctx, Z0 + row0 * c, pz, 8 * rows * c)); // straight into the caller's (messenger-facing) buffer
#undef G_TRY
        return release(0);
    });
}

int mpcx_group_aby3_matmul_local(mpcx_group *g, const int64_t *X0, const int64_t *X1, const int64_t *Y0,
                                 const int64_t *Y1, uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next,
                                 uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0) {
    if (!g) return group_fail(g, "mpcx_group_aby3_matmul_local: the group is NULL");
    if (!(X0 && X1 && Y0 && Y1 && Z0)) return group_fail(g, "mpcx_group_aby3_matmul_local: NULL operand");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_aby3_matmul_local: empty shape");
    if (layout != 0 && layout != 1) return group_fail(g, "mpcx_group_aby3_matmul_local: layout must be 0 or 1");
    return group_aby3_matmul(g, X0, X1, Y0, Y1, nullptr, r, l, c, layout, seed_next, pop_next, seed_prev, pop_prev, Z0);
}

int mpcx_group_aby3_matmul_local_resident(mpcx_group *g, const int64_t *X0, const int64_t *X1,
                                          const mpcx_group_operand *Y, uint64_t r, uint64_t l, uint64_t c, int layout,
                                          uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                                          int64_t *Z0) {
    if (!g) return group_fail(g, "mpcx_group_aby3_matmul_local_resident: the group is NULL");
    if (!(X0 && X1 && Y && Z0)) return group_fail(g, "mpcx_group_aby3_matmul_local_resident: NULL operand");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_aby3_matmul_local_resident: empty shape");
    if (layout != 0 && layout != 1) return group_fail(g, "mpcx_group_aby3_matmul_local_resident: layout must be 0 or 1");
    if (Y->rows != l || Y->cols != c || Y->s0.size() != g->ctx.size())
        return group_fail(g, "mpcx_group_aby3_matmul_local_resident: the operand is %llu x %llu on %zu members, the product "
                             "needs %llu x %llu on %zu", (unsigned long long)Y->rows, (unsigned long long)Y->cols,
                          Y->s0.size(), (unsigned long long)l, (unsigned long long)c, g->ctx.size());
    return group_aby3_matmul(g, X0, X1, nullptr, nullptr, Y, r, l, c, layout, seed_next, pop_next, seed_prev, pop_prev, Z0);
}

int mpcx_group_operand_upload(mpcx_group *g, const int64_t *M0, const int64_t *M1, uint64_t rows, uint64_t cols, int layout,
                              mpcx_group_operand **out) {
    if (!g || !out) return group_fail(g, "mpcx_group_operand_upload: NULL argument");
    if (!(M0 && M1 && rows && cols)) return group_fail(g, "mpcx_group_operand_upload: NULL or empty operand");
    if (layout != 0 && layout != 1) return group_fail(g, "mpcx_group_operand_upload: layout must be 0 or 1");
    const int members = (int)g->ctx.size();
    mpcx_group_operand *op = new mpcx_group_operand();
    op->rows = rows, op->cols = cols;
    op->s0.assign(members, nullptr);
    op->s1.assign(members, nullptr);
    op->prepared.assign(members, nullptr);
    const int rc = group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        void *tmp = nullptr;
        if (mpcx_malloc(ctx, 8 * rows * cols, &op->s0[k]) || mpcx_malloc(ctx, 8 * rows * cols, &op->s1[k])) return 1;
        int rc2 = 0;
        if (layout == 0) {
            rc2 = mpcx_h2d(ctx, op->s0[k], M0, 8 * rows * cols) || mpcx_h2d(ctx, op->s1[k], M1, 8 * rows * cols);
        } else {
            if (mpcx_malloc(ctx, 8 * rows * cols, &tmp)) return 1;
            rc2 = mpcx_h2d(ctx, tmp, M0, 8 * rows * cols);
            if (!rc2) rc2 = mpcx_relayout(ctx, (const i64 *)tmp, (i64 *)op->s0[k], rows, cols, 1, 0);
            if (!rc2) rc2 = mpcx_h2d(ctx, tmp, M1, 8 * rows * cols); // ordered behind the relayout on the context's stream
            if (!rc2) rc2 = mpcx_relayout(ctx, (const i64 *)tmp, (i64 *)op->s1[k], rows, cols, 1, 0);
            (void)mpcx_free(ctx, tmp);
        }
        if (!rc2)
            rc2 = mpcx_aby3_operand_prepare(ctx, (const i64 *)op->s0[k], (const i64 *)op->s1[k], rows, cols, 0, &op->prepared[k]);
        return rc2;
    });
    if (rc) {
        (void)mpcx_group_operand_free(g, op);
        return rc;
    }
    *out = op;
    return 0;
}

int mpcx_group_operand_free(mpcx_group *g, mpcx_group_operand *op) {
    if (!op) return 0;
    if (g)
        for (size_t k = 0; k < op->s0.size() && k < g->ctx.size(); k++) {
            (void)mpcx_aby3_operand_free(g->ctx[k], op->prepared[k]);
            (void)mpcx_free(g->ctx[k], op->s0[k]);
            (void)mpcx_free(g->ctx[k], op->s1[k]);
        }
    delete op;
    return 0;
}


// SecurennEngine::matrix_mul, parties 0 / 1 after round 2 (SecurennEngine.cpp:157-167): Z = C + E B + A F + party E F with
// E = E_own + E_peer, F = F_own + F_peer.  The rows of A, C, E (and so of Z) are spread over the members, B and F go to
// every member; a member's buffers are laid out like the messages of a product with its row count, so that the
// single-GPU segment runs on them unchanged.
int mpcx_group_snn_beaver_party_finish_matmul(mpcx_group *g, int party, const int64_t *triple, const int64_t *ef_own,
                                              const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int64_t *Z) {
    if (!g) return group_fail(g, "mpcx_group_snn_beaver_party_finish_matmul: the group is NULL");
    if (!(triple && ef_own && ef_peer && Z)) return group_fail(g, "mpcx_group_snn_beaver_party_finish_matmul: NULL operand");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_snn_beaver_party_finish_matmul: empty shape");
    if (party != 0 && party != 1) return group_fail(g, "mpcx_group_snn_beaver_party_finish_matmul: party %d", party);
    int members = (int)g->ctx.size();
    const u64 granule = 128;
    if (ceil_div(r, granule) < (u64)members) members = (int)ceil_div(r, granule);
    const u64 na = r * l, nb = l * c;
    return group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        u64 row0 = 0, rows = 0;
        mpcx_plan_split(r, members, k, granule, &row0, &rows);
        if (rows == 0) return 0;
        const u64 ma = rows * l, mc = rows * c;
        void *pt = nullptr, *po = nullptr, *pp = nullptr, *pz = nullptr;
        auto release = [&](int rc) {
            for (void *p : {pt, po, pp, pz}) (void)mpcx_free(ctx, p);
            return rc;
        };
#define G_TRY(call)                                                                                                    \
    do {                                                                                                               \
        if ((call) != 0) return release(1);                                                                            \
    } while (0)
        G_TRY(mpcx_malloc(ctx, 8 * (ma + nb + mc), &pt));
        G_TRY(mpcx_malloc(ctx, 8 * (ma + nb), &po));
        G_TRY(mpcx_malloc(ctx, 8 * (ma + nb), &pp));
        G_TRY(mpcx_malloc(ctx, 8 * mc, &pz));
        i64 *t = (i64 *)pt, *o = (i64 *)po, *q = (i64 *)pp;
        G_TRY(mpcx_h2d(ctx, t, triple + row0 * l, 8 * ma));                 // A rows
        G_TRY(mpcx_h2d(ctx, t + ma, triple + na, 8 * nb));                  // B
        G_TRY(mpcx_h2d(ctx, t + ma + nb, triple + na + nb + row0 * c, 8 * mc)); // C rows
        G_TRY(mpcx_h2d(ctx, o, ef_own + row0 * l, 8 * ma));
        G_TRY(mpcx_h2d(ctx, o + ma, ef_own + na, 8 * nb));
        G_TRY(mpcx_h2d(ctx, q, ef_peer + row0 * l, 8 * ma));
        G_TRY(mpcx_h2d(ctx, q + ma, ef_peer + na, 8 * nb));
        G_TRY(mpcx_snn_beaver_party_finish_matmul(ctx, party, t, o, q, rows, l, c, 0, (i64 *)pz));
        G_TRY(mpcx_d2h(ctx, Z + row0 * c, pz, 8 * mc));
#undef G_TRY
        return release(0);
    });
}


// SecurennEngine::matrix_mul, party 2 before round 1 (SecurennEngine.cpp:72-110): the dealer's product
// C_1 = (A_0 + A_1)(B_0 + B_1) - C_0 in row blocks; every member writes the rows it dealt into the two host messages.
int mpcx_group_snn_beaver_dealer_matmul(mpcx_group *g, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, int64_t *msg0,
                                        int64_t *msg1) {
    if (!g) return group_fail(g, "mpcx_group_snn_beaver_dealer_matmul: the group is NULL");
    if (!(msg0 && msg1)) return group_fail(g, "mpcx_group_snn_beaver_dealer_matmul: NULL pointer");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_snn_beaver_dealer_matmul: empty shape");
    int members = (int)g->ctx.size();
    const u64 granule = 128;
    if (ceil_div(r, granule) < (u64)members) members = (int)ceil_div(r, granule);
    const u64 na = r * l, nb = l * c;
    return group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        u64 row0 = 0, rows = 0;
        mpcx_plan_split(r, members, k, granule, &row0, &rows);
        if (rows == 0) return 0;
        const u64 ma = rows * l, mc = rows * c;
        const bool with_b = k == 0; // B_i travels once
        void *buf = nullptr;
        if (mpcx_malloc(ctx, 8 * (2 * ma + 2 * mc + (with_b ? 2 * nb : 0)), &buf) != 0) return 1;
        i64 *a0 = (i64 *)buf, *a1 = a0 + ma, *c0 = a1 + ma, *c1 = c0 + mc, *b0 = with_b ? c1 + mc : nullptr,
            *b1 = with_b ? b0 + nb : nullptr;
        int rc = mpcx_snn_beaver_dealer_matmul_rows(ctx, seedS, r, l, c, row0, rows, a0, a1, b0, b1, c0, c1);
        if (!rc) rc = mpcx_d2h(ctx, msg0 + row0 * l, a0, 8 * ma);
        if (!rc) rc = mpcx_d2h(ctx, msg1 + row0 * l, a1, 8 * ma);
        if (!rc) rc = mpcx_d2h(ctx, msg0 + na + nb + row0 * c, c0, 8 * mc);
        if (!rc) rc = mpcx_d2h(ctx, msg1 + na + nb + row0 * c, c1, 8 * mc);
        if (!rc && with_b) rc = mpcx_d2h(ctx, msg0 + na, b0, 8 * nb);
        if (!rc && with_b) rc = mpcx_d2h(ctx, msg1 + na, b1, 8 * nb);
        (void)mpcx_free(ctx, buf);
        return rc;
    });
}


// MyEngine::matrix_mul (MyEngine.cpp:145-219) in row blocks: the dealer's C_0 / C_1 rows land in the two host messages,
// the parties' Z rows in the caller's row-major buffer.
int mpcx_group_my_beaver_dealer_matmul(mpcx_group *g, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r,
                                       uint64_t l, uint64_t c, int64_t *C0, int64_t *C1) {
    if (!g) return group_fail(g, "mpcx_group_my_beaver_dealer_matmul: the group is NULL");
    if (!(C0 && C1)) return group_fail(g, "mpcx_group_my_beaver_dealer_matmul: NULL pointer");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_my_beaver_dealer_matmul: empty shape");
    int members = (int)g->ctx.size();
    const u64 granule = 128;
    if (ceil_div(r, granule) < (u64)members) members = (int)ceil_div(r, granule);
    return group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        u64 row0 = 0, rows = 0;
        mpcx_plan_split(r, members, k, granule, &row0, &rows);
        if (rows == 0) return 0;
        const u64 mc = rows * c;
        void *buf = nullptr;
        if (mpcx_malloc(ctx, 16 * mc, &buf) != 0) return 1;
        i64 *c0 = (i64 *)buf, *c1 = c0 + mc;
        int rc = mpcx_my_beaver_dealer_matmul_rows(ctx, seed0, seed1, seedS, r, l, c, row0, rows, c0, c1);
        if (!rc) rc = mpcx_d2h(ctx, C0 + row0 * c, c0, 8 * mc);
        if (!rc) rc = mpcx_d2h(ctx, C1 + row0 * c, c1, 8 * mc);
        (void)mpcx_free(ctx, buf);
        return rc;
    });
}

int mpcx_group_my_beaver_party_finish_matmul(mpcx_group *g, int party, uint64_t seed2, const int64_t *C,
                                             const int64_t *ef_own, const int64_t *ef_peer, uint64_t r, uint64_t l,
                                             uint64_t c, int64_t *Z) {
    if (!g) return group_fail(g, "mpcx_group_my_beaver_party_finish_matmul: the group is NULL");
    if (!(C && ef_own && ef_peer && Z)) return group_fail(g, "mpcx_group_my_beaver_party_finish_matmul: NULL operand");
    if (!(r && l && c)) return group_fail(g, "mpcx_group_my_beaver_party_finish_matmul: empty shape");
    if (party != 0 && party != 1) return group_fail(g, "mpcx_group_my_beaver_party_finish_matmul: party %d", party);
    int members = (int)g->ctx.size();
    const u64 granule = 128;
    if (ceil_div(r, granule) < (u64)members) members = (int)ceil_div(r, granule);
    const u64 na = r * l, nb = l * c;
    return group_run(g, members, [=](int k) -> int {
        mpcx_ctx *ctx = g->ctx[k];
        u64 row0 = 0, rows = 0;
        mpcx_plan_split(r, members, k, granule, &row0, &rows);
        if (rows == 0) return 0;
        const u64 ma = rows * l, mc = rows * c;
        void *buf = nullptr;
        if (mpcx_malloc(ctx, 8 * (2 * mc + 2 * (ma + nb)), &buf) != 0) return 1;
        i64 *cb = (i64 *)buf, *z = cb + mc, *o = z + mc, *q = o + ma + nb;
        int rc = mpcx_h2d(ctx, cb, C + row0 * c, 8 * mc);
        if (!rc) rc = mpcx_h2d(ctx, o, ef_own + row0 * l, 8 * ma);
        if (!rc) rc = mpcx_h2d(ctx, o + ma, ef_own + na, 8 * nb);
        if (!rc) rc = mpcx_h2d(ctx, q, ef_peer + row0 * l, 8 * ma);
        if (!rc) rc = mpcx_h2d(ctx, q + ma, ef_peer + na, 8 * nb);
        if (!rc) rc = mpcx_my_beaver_party_finish_matmul_rows(ctx, party, seed2, cb, o, q, r, l, c, row0, rows, 0, z);
        if (!rc) rc = mpcx_d2h(ctx, Z + row0 * c, z, 8 * mc);
        (void)mpcx_free(ctx, buf);
        return rc;
    });
}

} // extern "C"
