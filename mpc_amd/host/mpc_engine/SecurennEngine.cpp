#include "SecurennEngine.hpp"

#include <memory>

#include "../utils/util.hpp"
#include "gpu_support.hpp"

using mpcgpu::Buf;
using mpcgpu::Bufs;
using mpcgpu::Dev;
using mpcgpu::fresh;
using mpcgpu::gather;
using mpcgpu::get;
using mpcgpu::get_sections;
using mpcgpu::members;
using mpcgpu::put;
using mpcgpu::put_sections;
using mpcgpu::run;
using mpcgpu::scatter;
using mpcgpu::upload_string;

namespace {
const size_t W = sizeof(int64_t);
const size_t ZP = sizeof(zpshare); // 128 bytes per element
inline void append(string &dst, const string &src) { dst.append(src); }
void vector_mul_impl(SecurennEngine &e, const vector<int64_t> &Xj, const vector<int64_t> &Yj, vector<int64_t> &Zj);
} // namespace

SecurennEngine::SecurennEngine(VirtuleMessenger *_messenger, size_t _party_id)
    : messenger(_messenger), party_id(_party_id), prgs() {
    message_send.resize(3);
    message_recv.resize(3);
    static const uint64_t pair_key[3][3] = {{KEY0, KEY01, KEY20}, {KEY01, KEY1, KEY12}, {KEY20, KEY12, KEY2}};
    assert(party_id < 3);
    for (size_t k = 0; k < 3; k++) prgs.push_back(PRgenerator(pair_key[party_id][k], 0));
    group = mpcgpu::open_group();
    gpu = group ? mpcx_group_ctx(group, 0) : mpcgpu::open_context();
}

// a share matrix kept on the engine's first GPU, column-major as Eigen stores it
struct SecurennEngine::Pinned {
    const MatrixXl *key;
    size_t rows, cols;
    Dev<int64_t> d;
    Pinned(mpcx_ctx *ctx, const MatrixXl &M) : key(&M), rows(M.rows()), cols(M.cols()), d(ctx, M.data(), M.rows() * M.cols()) {}
};
SecurennEngine::Pinned *SecurennEngine::find_pinned(const MatrixXl &M) const {
    for (Pinned *p : pinned)
        if (p->key == &M) return p;
    return nullptr;
}
void SecurennEngine::pin(const MatrixXl &M) {
    unpin(M);
    if (party_id != 2) pinned.push_back(new Pinned(gpu, M)); // party 2 holds no share
}
void SecurennEngine::unpin(const MatrixXl &M) {
    for (size_t i = 0; i < pinned.size(); i++)
        if (pinned[i]->key == &M) {
            delete pinned[i];
            pinned.erase(pinned.begin() + i);
            return;
        }
}

SecurennEngine::~SecurennEngine() {
    for (Pinned *p : pinned) delete p;
    pinned.clear();
    if (group)
        mpcx_group_shutdown(group);
    else
        mpcx_shutdown(gpu);
}

// ---- Beaver-triple matrix product: party 2 deals (A_i | B_i | C_i), parties 0/1 open E = X - A, F = Y - B
void SecurennEngine::matrix_mul(const MatrixXl &Xj, const MatrixXl &Yj, MatrixXl &Zj) {
#if defined MYDEBUG // the product of the opened operands (the statement SecurennEngine.cpp:218 makes for vector_mul)
    mpcgpu::AtExit check([&] {
        MatrixXl xd, yd, zd;
        reveal_matrix(Xj, xd);
        reveal_matrix(Yj, yd);
        reveal_matrix(Zj, zd);
        assert(zd == xd * yd);
    });
#endif
    const size_t rows = Xj.rows(), lens = Xj.cols(), cols = Yj.cols();
    assert(Yj.rows() == lens);
    const size_t na = rows * lens, nb = lens * cols, nc = rows * cols;
    if (party_id == 2) {
        if (group && rows && lens && cols) { // the dealer's product in row blocks, dealt straight into the two messages
            message_send[0].resize((na + nb + nc) * W);
            message_send[1].resize((na + nb + nc) * W);
            if (mpcx_group_snn_beaver_dealer_matmul(group, prgs[2].init_seed, rows, lens, cols, (int64_t *)&message_send[0][0],
                                                    (int64_t *)&message_send[1][0]) != 0) {
                fprintf(stderr, "mpcx_group_snn_beaver_dealer_matmul: %s\n", mpcx_group_last_error(group));
                abort();
            }
        } else {
            Dev<int64_t> m0(gpu, na + nb + nc), m1(gpu, na + nb + nc);
            GPU_OK(gpu, mpcx_snn_beaver_dealer_matmul(gpu, prgs[2].init_seed, rows, lens, cols, m0, m1));
            m0.to_string(message_send[0]);
            m1.to_string(message_send[1]);
        }
        messenger->send_and_recv(message_send, message_recv); // round 1
        message_send[0].clear();
        message_send[1].clear();
        messenger->send_and_recv(message_send, message_recv); // round 2
        Zj.resize(rows, cols);
        return;
    }
    const size_t peer = 1 - party_id;
    for (auto &m : message_send) m.clear();
    messenger->send_and_recv(message_send, message_recv); // round 1
    MSG_CHECK(message_recv[2].size() == (na + nb + nc) * W);
    Dev<int64_t> triple(gpu, na + nb + nc), ef(gpu, na + nb), ef_peer(gpu, na + nb);
    upload_string(triple, message_recv[2]);
    string triple_msg; // round 2 overwrites the slot; a group reads the triple from the host once more
    if (group) triple_msg.swap(message_recv[2]);
    {
        const Pinned *px = find_pinned(Xj), *py = find_pinned(Yj); // operands the caller pinned are on the device already
        assert(!px || (px->rows == rows && px->cols == lens));
        assert(!py || (py->rows == lens && py->cols == cols));
        std::unique_ptr<Dev<int64_t>> ux, uy;
        if (!px) ux.reset(new Dev<int64_t>(gpu, Xj.data(), na));
        if (!py) uy.reset(new Dev<int64_t>(gpu, Yj.data(), nb));
        GPU_OK(gpu, mpcx_snn_beaver_party_open_matmul(gpu, px ? px->d.p : ux->p, py ? py->d.p : uy->p, 1, triple, rows, lens,
                                                      cols, ef));
    }
    ef.to_string(message_send[peer]);
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv); // round 2
    MSG_CHECK(message_recv[peer].size() == (na + nb) * W);
    Zj.resize(rows, cols);
    if (group && rows && lens && cols) {
        // row blocks over the group's GPUs, straight from the three messages; the product comes back row-major
        VecI64 zr(nc);
        if (mpcx_group_snn_beaver_party_finish_matmul(group, (int)party_id, (const int64_t *)triple_msg.data(),
                                                      (const int64_t *)message_send[peer].data(),
                                                      (const int64_t *)message_recv[peer].data(), rows, lens, cols,
                                                      zr.data()) != 0) {
            fprintf(stderr, "mpcx_group_snn_beaver_party_finish_matmul: %s\n", mpcx_group_last_error(group));
            abort();
        }
        vecI642matrix(zr, Zj);
        return;
    }
    upload_string(ef_peer, message_recv[peer]);
    Dev<int64_t> z(gpu, nc);
    GPU_OK(gpu, mpcx_snn_beaver_party_finish_matmul(gpu, (int)party_id, triple, ef, ef_peer, rows, lens, cols, 1, z));
    z.down(Zj.data(), nc);
}

void SecurennEngine::vector_mul(const vector<int64_t> &Xj, const vector<int64_t> &Yj, vector<int64_t> &Zj) {
#if defined MYDEBUG // :218, :296
    mpcgpu::AtExit check([&] {
        vector<int64_t> xd, yd, zd;
        reveal(Xj, xd);
        reveal(Yj, yd);
        reveal(Zj, zd);
        for (size_t i = 0; i < zd.size(); i++) assert((uint64_t)zd[i] == (uint64_t)xd[i] * (uint64_t)yd[i]);
    });
#endif
    const size_t n = Xj.size();
    assert(Yj.size() == n);
    Zj.assign(n, 0);
    vector_mul_impl(*this, Xj, Yj, Zj);
}

// ---- additive sharing between parties 0 and 1 out of their common stream
void SecurennEngine::share(const vector<int64_t> &X_plain, vector<int64_t> &Xj) {
    const size_t n = X_plain.size();
    Xj.assign(n, 0);
    if (party_id == 2 || n == 0) return;
    PRgenerator &g = prgs[1 - party_id];
    Dev<int64_t> x(gpu, X_plain.data(), n), out(gpu, n);
    GPU_OK(gpu, mpcx_snn_share_local(gpu, (int)party_id, x, n, g.init_seed, g.pops, out));
    g.advance(gpu, n);
    out.down(Xj.data(), n);
}

// ---- openings: parties 0 and 1 swap shares and copy them to party 2
void SecurennEngine::reveal(const vector<int64_t> &Xj, vector<int64_t> &X_plain) {
    const size_t n = Xj.size();
    X_plain.assign(n, 0);
    const size_t a = party_id == 2 ? 0 : 1 - party_id, b = party_id == 2 ? 1 : party_id;
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vector2string(Xj, message_send[1 - party_id]);
        message_send[2] = message_send[1 - party_id];
    }
    messenger->send_and_recv(message_send, message_recv);
    Dev<int64_t> u(gpu, n), v(gpu, n), out(gpu, n);
    MSG_CHECK(message_recv[a].size() == n * W);
    upload_string(u, message_recv[a]);
    if (party_id == 2) {
        MSG_CHECK(message_recv[1].size() == n * W);
        upload_string(v, message_recv[b]);
    } else {
        v.up(Xj.data(), n);
    }
    GPU_OK(gpu, mpcx_ring_add(gpu, u, v, out, n));
    out.down(X_plain.data(), n);
}

void SecurennEngine::reveal_matrix(const MatrixXl &Xj, MatrixXl &X_plain) {
    const size_t rows = Xj.rows(), cols = Xj.cols(), n = rows * cols;
    X_plain.resize(rows, cols);
    string own;
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        matrix2string(Xj, own);
        message_send[1 - party_id] = own;
        message_send[2] = own;
    }
    messenger->send_and_recv(message_send, message_recv);
    const string &u_s = party_id == 2 ? message_recv[0] : message_recv[1 - party_id];
    const string &v_s = party_id == 2 ? message_recv[1] : own;
    assert(u_s.size() == n * W && v_s.size() == n * W);
    Dev<int64_t> u(gpu, n), v(gpu, n), out(gpu, n);
    upload_string(u, u_s);
    upload_string(v, v_s);
    GPU_OK(gpu, mpcx_ring_add(gpu, u, v, out, n));
    string res;
    out.to_string(res);
    string2matrix(res, X_plain);
}

void SecurennEngine::reveal_vec_bools(const vector<int8_t> &Xj, vector<int8_t> &X_plain) {
    const size_t n = Xj.size();
    X_plain.assign(n, 0);
    if (party_id == 2) {
        vecbool2string(Xj, message_send[0]);
        vecbool2string(Xj, message_send[1]);
    } else {
        vecbool2string(Xj, message_send[2]);
        message_send[1 - party_id].clear();
    }
    messenger->send_and_recv(message_send, message_recv);
    const string &other = party_id == 2 ? message_recv[0] : message_recv[2];
    if (party_id == 2) assert(message_recv[0] == message_recv[1]);
    assert(other.size() == n);
    vector<int8_t> zeros(n, 0);
    Dev<int8_t> a(gpu, Xj.data(), n), b(gpu, n), c(gpu, zeros.data(), n), out(gpu, n);
    upload_string(b, other);
    GPU_OK(gpu, mpcx_aby3_reveal3_bools(gpu, a, b, c, out, n));
    out.down(X_plain.data(), n);
}

void SecurennEngine::reveal_L_1(const vector<int64_t> &Xj, vector<int64_t> &X_plain) {
    const size_t n = Xj.size();
    X_plain.assign(n, 0);
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vector2string(Xj, message_send[1 - party_id]);
        message_send[2] = message_send[1 - party_id];
    }
    messenger->send_and_recv(message_send, message_recv);
    Dev<int64_t> u(gpu, n), v(gpu, n), out(gpu, n);
    if (party_id == 2) {
        MSG_CHECK(message_recv[0].size() == n * W && message_recv[1].size() == n * W);
        upload_string(u, message_recv[0]);
        upload_string(v, message_recv[1]);
    } else {
        MSG_CHECK(message_recv[1 - party_id].size() == n * W);
        v.up(Xj.data(), n);
        upload_string(u, message_recv[1 - party_id]);
    }
    // add_L_1(Xj, received) on parties 0/1, add_L_1(X0, X1) on party 2 (the operation is commutative)
    GPU_OK(gpu, mpcx_snn_add_L1(gpu, v, u, out, n));
    out.down(X_plain.data(), n);
}

void SecurennEngine::reveal_zp(const vector<zpshare> &Xbits_j, vector<zpshare> &Xbits_plain) {
    const size_t n = Xbits_j.size();
    Xbits_plain.resize(n);
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vecZpshare2string(Xbits_j, message_send[1 - party_id]);
        message_send[2] = message_send[1 - party_id];
    }
    messenger->send_and_recv(message_send, message_recv);
    Dev<uint16_t> u(gpu, 64 * n), v(gpu, 64 * n), out(gpu, 64 * n);
    if (party_id == 2) {
        MSG_CHECK(message_recv[0].size() == n * ZP && message_recv[1].size() == n * ZP);
        upload_string(u, message_recv[0]);
        upload_string(v, message_recv[1]);
    } else {
        MSG_CHECK(message_recv[1 - party_id].size() == n * ZP);
        upload_string(u, message_recv[1 - party_id]);
        v.up(Xbits_j.data(), 64 * n);
    }
    GPU_OK(gpu, mpcx_snn_add_zp(gpu, u, v, out, 64 * n));
    out.down(Xbits_plain.data(), 64 * n);
}

// The multi-round protocols below keep every intermediate value on the device: a message is downloaded once
// into its send slot and a received message uploaded once; the public methods only move their arguments in and
// their results out.  (`e` is the engine; these helpers are file-local so the class keeps the reference's surface.)
//
// Every protocol is written over element ranges: run() hands the whole vector to the engine's one GPU, or -- with
// MPCX_DEVICES -- every member of the group its own contiguous range [first, first + count) on the member's thread.
// A member keeps its device buffers across the rounds (one slot per member in the vectors below), runs the range
// forms (n_total, i0, n) of the library's segments and moves its own slices of the messages: a message is the
// concatenation of sections (one per array it carries), each section holding the elements in order.
namespace {

struct PcIn { // what a party brings to privateCompare, per member
    Bufs<uint16_t> xbits;
    Bufs<int8_t> beta;
    Bufs<int64_t> r;
    explicit PcIn(size_t m) : xbits(m), beta(m), r(m) {}
};

// privateCompare, parties 0/1 (one round): d -> party 2
void pc_party_dev(SecurennEngine &e, const PcIn &in, size_t n) {
    const uint64_t s01 = e.prgs[1 - e.party_id].init_seed;
    e.message_send[1 - e.party_id].clear();
    string &out = e.message_send[2];
    out.resize(n * ZP);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<uint16_t> d(ctx, 64 * cnt);
        GPU_OK(ctx, mpcx_snn_pc_party(ctx, (int)e.party_id, in.xbits[m]->p, in.beta[m]->p, in.r[m]->p, n, i0, cnt, s01, d));
        put(d, out, 0, i0, cnt, 64);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv);
}
// privateCompare, party 2 (one round): beta' on the device
void pc_helper_dev(SecurennEngine &e, size_t n, Bufs<int8_t> &beta2) {
    e.message_send[0].clear();
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv);
    MSG_CHECK(e.message_recv[0].size() == n * ZP && e.message_recv[1].size() == n * ZP);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<uint16_t> d0(ctx, 64 * cnt), d1(ctx, 64 * cnt);
        get(d0, e.message_recv[0], 0, i0, cnt, 64);
        get(d1, e.message_recv[1], 0, i0, cnt, 64);
        GPU_OK(ctx, mpcx_snn_pc_helper(ctx, d0, d1, cnt, fresh(beta2[m], ctx, cnt)));
    });
}
// vector_mul, party 2 (two rounds): A_i | B_i | C_i to party i
void vecmul_dealer(SecurennEngine &e, size_t n) {
    for (size_t j = 0; j < 2; j++) e.message_send[j].resize(3 * n * W);
    run(e, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> t0(ctx, 3 * cnt), t1(ctx, 3 * cnt);
        GPU_OK(ctx, mpcx_snn_beaver_dealer_vecmul(ctx, e.prgs[2].init_seed, n, i0, cnt, t0.p, t0.p + cnt, t0.p + 2 * cnt, t1.p,
                                                  t1.p + cnt, t1.p + 2 * cnt));
        put_sections(t0, e.message_send[0], 0, 3, n, i0, cnt);
        put_sections(t1, e.message_send[1], 0, 3, n, i0, cnt);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    e.message_send[0].clear();
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
}
// vector_mul, parties 0/1 (two rounds): z = share of x * y
void vecmul_party_dev(SecurennEngine &e, const Bufs<int64_t> &x, const Bufs<int64_t> &y, size_t n, Bufs<int64_t> &z) {
    const size_t peer = 1 - e.party_id, G = members(e);
    for (auto &m : e.message_send) m.clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    MSG_CHECK(e.message_recv[2].size() == 3 * n * W);
    Bufs<int64_t> t(G), ef(G);
    string &out = e.message_send[peer];
    out.resize(2 * n * W);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> &tm = fresh(t[m], ctx, 3 * cnt), &efm = fresh(ef[m], ctx, 2 * cnt);
        get_sections(tm, e.message_recv[2], 0, 3, n, i0, cnt);
        GPU_OK(ctx, mpcx_snn_beaver_party_open_vecmul(ctx, x[m]->p, y[m]->p, tm.p, tm.p + cnt, cnt, efm.p, efm.p + cnt));
        put_sections(efm, out, 0, 2, n, i0, cnt);
    });
    e.message_send[2].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
    MSG_CHECK(e.message_recv[peer].size() == 2 * n * W);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> efp(ctx, 2 * cnt);
        get_sections(efp, e.message_recv[peer], 0, 2, n, i0, cnt);
        Dev<int64_t> &tm = *t[m], &efm = *ef[m];
        GPU_OK(ctx, mpcx_snn_beaver_party_finish_vecmul(ctx, (int)e.party_id, tm.p, tm.p + cnt, tm.p + 2 * cnt, efm.p,
                                                        efm.p + cnt, efp.p, efp.p + cnt, cnt, fresh(z[m], ctx, cnt)));
    });
}

// shareConvert (four rounds); parties 0/1: a -> y, party 2: no input, no output
void share_convert_dev(SecurennEngine &e, const Bufs<int64_t> *a, size_t n, Bufs<int64_t> *y) {
    const size_t G = members(e);
    if (e.party_id != 2) {
        const size_t peer = 1 - e.party_id;
        const uint64_t s01 = e.prgs[peer].init_seed;
        e.message_send[peer].clear();
        e.message_send[2].resize(n * W);
        run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> x(ctx, cnt);
            GPU_OK(ctx, mpcx_snn_sc_party_open(ctx, (int)e.party_id, (*a)[m]->p, n, i0, cnt, s01, x));
            put(x, e.message_send[2], 0, i0, cnt);
        });
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
        e.message_send[2].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2: x bits over Z_67 | delta_j
        MSG_CHECK(e.message_recv[2].size() == n * (ZP + W));
        PcIn pc(G);
        Bufs<int64_t> delta(G);
        run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            get(fresh(pc.xbits[m], ctx, 64 * cnt), e.message_recv[2], 0, i0, cnt, 64);
            get(fresh(delta[m], ctx, cnt), e.message_recv[2], n * ZP, i0, cnt);
            GPU_OK(ctx, mpcx_snn_sc_party_pc_inputs(ctx, n, i0, cnt, s01, fresh(pc.beta[m], ctx, cnt), fresh(pc.r[m], ctx, cnt)));
        });
        pc_party_dev(e, pc, n); // round 3
        e.message_send[peer].clear();
        e.message_send[2].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4: eta'_j
        MSG_CHECK(e.message_recv[2].size() == n * W);
        run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> eta1(ctx, cnt);
            get(eta1, e.message_recv[2], 0, i0, cnt);
            GPU_OK(ctx, mpcx_snn_sc_party_finish(ctx, (int)e.party_id, (*a)[m]->p, delta[m]->p, eta1, n, i0, cnt, s01,
                                                 fresh((*y)[m], ctx, cnt)));
        });
        return;
    }
    const uint64_t sS = e.prgs[2].init_seed;
    e.message_send[0].clear(); // nothing to say in round 1 (parties 0/1 ignore this slot)
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    MSG_CHECK(e.message_recv[0].size() == n * W && e.message_recv[1].size() == n * W);
    for (size_t j = 0; j < 2; j++) e.message_send[j].resize(n * (ZP + W)); // bit shares | delta_j
    run(e, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> x0(ctx, cnt), x1(ctx, cnt), d0(ctx, cnt), d1(ctx, cnt);
        Dev<uint16_t> xb0(ctx, 64 * cnt), xb1(ctx, 64 * cnt);
        get(x0, e.message_recv[0], 0, i0, cnt);
        get(x1, e.message_recv[1], 0, i0, cnt);
        GPU_OK(ctx, mpcx_snn_sc_helper_split(ctx, x0, x1, n, i0, cnt, sS, xb0, d0, xb1, d1));
        put(xb0, e.message_send[0], 0, i0, cnt, 64);
        put(d0, e.message_send[0], n * ZP, i0, cnt);
        put(xb1, e.message_send[1], 0, i0, cnt, 64);
        put(d1, e.message_send[1], n * ZP, i0, cnt);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
    Bufs<int8_t> eta1_plain(G);
    pc_helper_dev(e, n, eta1_plain); // round 3
    for (size_t j = 0; j < 2; j++) e.message_send[j].resize(n * W);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> e0(ctx, cnt), e1(ctx, cnt);
        GPU_OK(ctx, mpcx_snn_sc_helper_eta(ctx, eta1_plain[m]->p, n, i0, cnt, sS, e0, e1));
        put(e0, e.message_send[0], 0, i0, cnt);
        put(e1, e.message_send[1], 0, i0, cnt);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4
}

// msb over Z_{2^64-1} (= lsb of 2a), result shared over Z_2^64; six rounds
void msb_L1_dev(SecurennEngine &e, const Bufs<int64_t> *a, size_t n, Bufs<int64_t> *b) {
    const size_t G = members(e);
    if (e.party_id == 2) {
        const uint64_t sS = e.prgs[2].init_seed;
        for (size_t j = 0; j < 2; j++) e.message_send[j].resize(n * (2 * W + ZP)); // x_j | lsb(x)_j | bits of x over Z_67
        run(e, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> x0(ctx, cnt), l0(ctx, cnt), x1(ctx, cnt), l1(ctx, cnt);
            Dev<uint16_t> xb0(ctx, 64 * cnt), xb1(ctx, 64 * cnt);
            GPU_OK(ctx, mpcx_snn_msbl1_helper_deal(ctx, n, i0, cnt, sS, x0, l0, xb0, x1, l1, xb1));
            put(x0, e.message_send[0], 0, i0, cnt);
            put(l0, e.message_send[0], n * W, i0, cnt);
            put(xb0, e.message_send[0], 2 * n * W, i0, cnt, 64);
            put(x1, e.message_send[1], 0, i0, cnt);
            put(l1, e.message_send[1], n * W, i0, cnt);
            put(xb1, e.message_send[1], 2 * n * W, i0, cnt, 64);
        });
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
        e.message_send[0].clear();
        e.message_send[1].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
        Bufs<int8_t> beta1_plain(G);
        pc_helper_dev(e, n, beta1_plain); // round 3
        for (size_t j = 0; j < 2; j++) e.message_send[j].resize(n * W);
        run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> b0(ctx, cnt), b1(ctx, cnt);
            GPU_OK(ctx, mpcx_snn_msbl1_helper_beta(ctx, beta1_plain[m]->p, n, i0, cnt, sS, b0, b1));
            put(b0, e.message_send[0], 0, i0, cnt);
            put(b1, e.message_send[1], 0, i0, cnt);
        });
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4
        vecmul_dealer(e, n);                                        // rounds 5-6
        return;
    }
    const size_t peer = 1 - e.party_id;
    const uint64_t s01 = e.prgs[peer].init_seed;
    e.message_send[2].clear();
    e.message_send[peer].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    MSG_CHECK(e.message_recv[2].size() == n * (2 * W + ZP));
    PcIn pc(G);
    Bufs<int64_t> xlsb(G), rj(G);
    e.message_send[peer].resize(n * W);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> xj(ctx, cnt);
        get(xj, e.message_recv[2], 0, i0, cnt);
        get(fresh(xlsb[m], ctx, cnt), e.message_recv[2], n * W, i0, cnt);
        get(fresh(pc.xbits[m], ctx, 64 * cnt), e.message_recv[2], 2 * n * W, i0, cnt, 64);
        GPU_OK(ctx, mpcx_snn_msbl1_party_open(ctx, (*a)[m]->p, xj, cnt, fresh(rj[m], ctx, cnt)));
        put(*rj[m], e.message_send[peer], 0, i0, cnt);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2: open r = 2a + x
    MSG_CHECK(e.message_recv[peer].size() == n * W);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> rpeer(ctx, cnt);
        get(rpeer, e.message_recv[peer], 0, i0, cnt);
        GPU_OK(ctx, mpcx_snn_msbl1_party_pc_inputs(ctx, rj[m]->p, rpeer, i0, cnt, s01, fresh(pc.beta[m], ctx, cnt),
                                                   fresh(pc.r[m], ctx, cnt)));
    });
    pc_party_dev(e, pc, n); // round 3
    e.message_send[2].clear();
    e.message_send[peer].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4: share of beta'
    MSG_CHECK(e.message_recv[2].size() == n * W);
    Bufs<int64_t> gamma(G), delta(G), theta(G);
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> beta1(ctx, cnt);
        get(beta1, e.message_recv[2], 0, i0, cnt);
        GPU_OK(ctx, mpcx_snn_msbl1_party_mid(ctx, (int)e.party_id, beta1, xlsb[m]->p, pc.beta[m]->p, pc.r[m]->p, cnt,
                                             fresh(gamma[m], ctx, cnt), fresh(delta[m], ctx, cnt)));
    });
    vecmul_party_dev(e, gamma, delta, n, theta); // rounds 5-6
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t, uint64_t cnt) {
        GPU_OK(ctx, mpcx_snn_msbl1_party_finish(ctx, gamma[m]->p, delta[m]->p, theta[m]->p, cnt, fresh((*b)[m], ctx, cnt)));
    });
}

void vector_mul_impl(SecurennEngine &e, const vector<int64_t> &Xj, const vector<int64_t> &Yj, vector<int64_t> &Zj) {
    const size_t n = Xj.size(), G = members(e);
    if (e.party_id == 2) return vecmul_dealer(e, n);
    Bufs<int64_t> x(G), y(G), z(G);
    scatter(e, Xj.data(), n, 1, x);
    scatter(e, Yj.data(), n, 1, y);
    vecmul_party_dev(e, x, y, n, z);
    gather(e, z, n, Zj.data());
}

} // namespace

// ---- privateCompare: parties 0/1 send masked difference digits to party 2, which learns beta' = beta xor (x > r)
void SecurennEngine::privateCompare(const vector<zpshare> &Xbits_j, const vector<int8_t> &Beta1_plain,
                                    const vector<int64_t> &R_palin, vector<int8_t> &Beta2_plain) {
    const size_t n = Xbits_j.size(), G = members(*this);
    assert(Beta1_plain.size() == n && R_palin.size() == n);
    Beta2_plain.assign(n, 0);
    if (party_id != 2) {
        PcIn pc(G);
        scatter(*this, (const uint16_t *)Xbits_j.data(), n, 64, pc.xbits);
        scatter(*this, Beta1_plain.data(), n, 1, pc.beta);
        scatter(*this, R_palin.data(), n, 1, pc.r);
        pc_party_dev(*this, pc, n);
        return;
    }
    Bufs<int8_t> b(G);
    pc_helper_dev(*this, n, b);
    gather(*this, b, n, Beta2_plain.data());
}

// ---- shareConvert: shares over Z_2^64 -> shares over Z_{2^64-1}, four rounds
void SecurennEngine::shareConvert(const vector<int64_t> &Aj, vector<int64_t> &Yj) {
#if defined MYDEBUG // :767, :876: the same value, now shared over Z_(2^64 - 1)
    mpcgpu::AtExit check([&] {
        vector<int64_t> ad, yd;
        reveal(Aj, ad);
        reveal_L_1(Yj, yd);
        for (size_t i = 0; i < ad.size(); i++) assert((uint64_t)yd[i] == (uint64_t)ad[i] % L_1);
    });
#endif
    const size_t n = Aj.size(), G = members(*this);
    Yj.assign(n, 0);
    if (party_id == 2) {
        share_convert_dev(*this, nullptr, n, nullptr);
        return;
    }
    Bufs<int64_t> a(G), y(G);
    scatter(*this, Aj.data(), n, 1, a);
    share_convert_dev(*this, &a, n, &y);
    gather(*this, y, n, Yj.data());
}

// ---- msb over Z_{2^64-1}, six rounds
void SecurennEngine::msb_L_1(const vector<int64_t> &Aj, vector<int64_t> &Bj) {
#if defined MYDEBUG // :961, :1038: the top bit of the value shared over Z_(2^64 - 1)
    mpcgpu::AtExit check([&] {
        vector<int64_t> ad, bd;
        reveal_L_1(Aj, ad);
        reveal(Bj, bd);
        for (size_t i = 0; i < ad.size(); i++) assert((uint64_t)bd[i] == (uint64_t)ad[i] >> 63);
    });
#endif
    const size_t n = Aj.size(), G = members(*this);
    Bj.assign(n, 0);
    if (party_id == 2) {
        msb_L1_dev(*this, nullptr, n, nullptr);
        return;
    }
    Bufs<int64_t> a(G), b(G);
    scatter(*this, Aj.data(), n, 1, a);
    msb_L1_dev(*this, &a, n, &b);
    gather(*this, b, n, Bj.data());
}

// ---- msb of a in [-2^62, 2^62): convert 2a to Z_{2^64-1}, then msb_L_1; the intermediate stays on the device
void SecurennEngine::msb(const vector<int64_t> &Aj, vector<int64_t> &Bj) {
#if defined MYDEBUG // :1102 asserts the input range; the sign bit comes out
    mpcgpu::AtExit check([&] {
        vector<int64_t> ad, bd;
        reveal(Aj, ad);
        reveal(Bj, bd);
        for (size_t i = 0; i < ad.size(); i++) {
            assert(-((int64_t)1 << 62) <= ad[i] && ad[i] <= ((int64_t)1 << 62) - 1);
            assert(bd[i] == (ad[i] < 0));
        }
    });
#endif
    const size_t n = Aj.size(), G = members(*this);
    Bj.assign(n, 0);
    if (party_id == 2) {
        share_convert_dev(*this, nullptr, n, nullptr);
        msb_L1_dev(*this, nullptr, n, nullptr);
        return;
    }
    Bufs<int64_t> a(G), twoA(G), y(G), b(G);
    scatter(*this, Aj.data(), n, 1, a);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t, uint64_t cnt) {
        GPU_OK(ctx, mpcx_ring_shl1(ctx, a[m]->p, fresh(twoA[m], ctx, cnt), cnt));
    });
    share_convert_dev(*this, &twoA, n, &y);
    msb_L1_dev(*this, &y, n, &b);
    gather(*this, b, n, Bj.data());
}

// ---- ReLU(a) = a - a * [a < 0] as one protocol: msb (ten rounds), vector_mul (two rounds), subtraction
void SecurennEngine::relu(const vector<int64_t> &Aj, vector<int64_t> &Rj) {
#if defined MYDEBUG // the opened result is max(a, 0)
    mpcgpu::AtExit check([&] {
        vector<int64_t> ad, rd;
        reveal(Aj, ad);
        reveal(Rj, rd);
        for (size_t i = 0; i < ad.size(); i++) assert(rd[i] == (ad[i] < 0 ? 0 : ad[i]));
    });
#endif
    const size_t n = Aj.size(), G = members(*this);
    Rj.assign(n, 0);
    if (party_id == 2) {
        share_convert_dev(*this, nullptr, n, nullptr);
        msb_L1_dev(*this, nullptr, n, nullptr);
        vecmul_dealer(*this, n);
        return;
    }
    Bufs<int64_t> a(G), twoA(G), y(G), b(G), z(G), r(G);
    scatter(*this, Aj.data(), n, 1, a);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t, uint64_t cnt) {
        GPU_OK(ctx, mpcx_ring_shl1(ctx, a[m]->p, fresh(twoA[m], ctx, cnt), cnt));
    });
    share_convert_dev(*this, &twoA, n, &y);
    msb_L1_dev(*this, &y, n, &b);
    vecmul_party_dev(*this, a, b, n, z);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t, uint64_t cnt) {
        GPU_OK(ctx, mpcx_ring_sub(ctx, a[m]->p, z[m]->p, fresh(r[m], ctx, cnt), cnt));
    });
    gather(*this, r, n, Rj.data());
}
