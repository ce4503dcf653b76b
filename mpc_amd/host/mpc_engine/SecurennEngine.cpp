#include "SecurennEngine.hpp"

#include "../utils/util.hpp"
#include "gpu_support.hpp"

using mpcgpu::Dev;
using mpcgpu::upload_string;

namespace {
const size_t W = sizeof(int64_t);
const size_t ZP = sizeof(zpshare); // 128 bytes per element
inline void append(string &dst, const string &src) { dst.append(src); }
} // namespace

SecurennEngine::SecurennEngine(VirtuleMessenger *_messenger, size_t _party_id)
    : messenger(_messenger), party_id(_party_id), prgs() {
    message_send.resize(3);
    message_recv.resize(3);
    static const uint64_t pair_key[3][3] = {{KEY0, KEY01, KEY20}, {KEY01, KEY1, KEY12}, {KEY20, KEY12, KEY2}};
    assert(party_id < 3);
    for (size_t k = 0; k < 3; k++) prgs.push_back(PRgenerator(pair_key[party_id][k], 0));
    gpu = mpcgpu::open_context();
}

SecurennEngine::~SecurennEngine() { mpcx_shutdown(gpu); }

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
        Dev<int64_t> m0(gpu, na + nb + nc), m1(gpu, na + nb + nc);
        GPU_OK(gpu, mpcx_snn_beaver_dealer_matmul(gpu, prgs[2].init_seed, rows, lens, cols, m0, m1));
        m0.to_string(message_send[0]);
        m1.to_string(message_send[1]);
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
    {
        Dev<int64_t> x(gpu, Xj.data(), na), y(gpu, Yj.data(), nb);
        GPU_OK(gpu, mpcx_snn_beaver_party_open_matmul(gpu, x, y, 1, triple, rows, lens, cols, ef));
    }
    ef.to_string(message_send[peer]);
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv); // round 2
    MSG_CHECK(message_recv[peer].size() == (na + nb) * W);
    upload_string(ef_peer, message_recv[peer]);
    Dev<int64_t> z(gpu, nc);
    GPU_OK(gpu, mpcx_snn_beaver_party_finish_matmul(gpu, (int)party_id, triple, ef, ef_peer, rows, lens, cols, 1, z));
    Zj.resize(rows, cols);
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
    if (party_id == 2) {
        Dev<int64_t> t0(gpu, 3 * n), t1(gpu, 3 * n); // A_i | B_i | C_i
        GPU_OK(gpu, mpcx_snn_beaver_dealer_vecmul(gpu, prgs[2].init_seed, n, 0, n, t0.p, t0.p + n, t0.p + 2 * n, t1.p,
                                                  t1.p + n, t1.p + 2 * n));
        t0.to_string(message_send[0]);
        t1.to_string(message_send[1]);
        messenger->send_and_recv(message_send, message_recv); // round 1
        message_send[0].clear();
        message_send[1].clear();
        messenger->send_and_recv(message_send, message_recv); // round 2
        return;
    }
    const size_t peer = 1 - party_id;
    for (auto &m : message_send) m.clear();
    messenger->send_and_recv(message_send, message_recv); // round 1
    MSG_CHECK(message_recv[2].size() == 3 * n * W);
    Dev<int64_t> t(gpu, 3 * n), ef(gpu, 2 * n), efp(gpu, 2 * n), z(gpu, n);
    upload_string(t, message_recv[2]);
    {
        Dev<int64_t> x(gpu, Xj.data(), n), y(gpu, Yj.data(), n);
        GPU_OK(gpu, mpcx_snn_beaver_party_open_vecmul(gpu, x, y, t.p, t.p + n, n, ef.p, ef.p + n));
    }
    ef.to_string(message_send[peer]);
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv); // round 2
    MSG_CHECK(message_recv[peer].size() == 2 * n * W);
    upload_string(efp, message_recv[peer]);
    GPU_OK(gpu, mpcx_snn_beaver_party_finish_vecmul(gpu, (int)party_id, t.p, t.p + n, t.p + 2 * n, ef.p, ef.p + n,
                                                    efp.p, efp.p + n, n, z));
    z.down(Zj.data(), n);
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
namespace {

// privateCompare, parties 0/1 (one round): d -> party 2
void pc_party_dev(SecurennEngine &e, const Dev<uint16_t> &xbits, const Dev<int8_t> &beta, const Dev<int64_t> &r, size_t n) {
    Dev<uint16_t> d(e.gpu, 64 * n);
    GPU_OK(e.gpu, mpcx_snn_pc_party(e.gpu, (int)e.party_id, xbits, beta, r, n, 0, n, e.prgs[1 - e.party_id].init_seed, d));
    e.message_send[1 - e.party_id].clear();
    d.to_string(e.message_send[2]);
    e.messenger->send_and_recv(e.message_send, e.message_recv);
}
// privateCompare, party 2 (one round): beta' on the device
void pc_helper_dev(SecurennEngine &e, size_t n, Dev<int8_t> &beta2) {
    e.message_send[0].clear();
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv);
    MSG_CHECK(e.message_recv[0].size() == n * ZP && e.message_recv[1].size() == n * ZP);
    Dev<uint16_t> d0(e.gpu, 64 * n), d1(e.gpu, 64 * n);
    upload_string(d0, e.message_recv[0]);
    upload_string(d1, e.message_recv[1]);
    GPU_OK(e.gpu, mpcx_snn_pc_helper(e.gpu, d0, d1, n, beta2));
}
// vector_mul, party 2 (two rounds)
void vecmul_dealer(SecurennEngine &e, size_t n) {
    Dev<int64_t> t0(e.gpu, 3 * n), t1(e.gpu, 3 * n); // A_i | B_i | C_i
    GPU_OK(e.gpu, mpcx_snn_beaver_dealer_vecmul(e.gpu, e.prgs[2].init_seed, n, 0, n, t0.p, t0.p + n, t0.p + 2 * n, t1.p,
                                                t1.p + n, t1.p + 2 * n));
    t0.to_string(e.message_send[0]);
    t1.to_string(e.message_send[1]);
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    e.message_send[0].clear();
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
}
// vector_mul, parties 0/1 (two rounds): z = share of x * y
void vecmul_party_dev(SecurennEngine &e, const Dev<int64_t> &x, const Dev<int64_t> &y, size_t n, Dev<int64_t> &z) {
    const size_t peer = 1 - e.party_id;
    for (auto &m : e.message_send) m.clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    MSG_CHECK(e.message_recv[2].size() == 3 * n * W);
    Dev<int64_t> t(e.gpu, 3 * n), ef(e.gpu, 2 * n), efp(e.gpu, 2 * n);
    upload_string(t, e.message_recv[2]);
    GPU_OK(e.gpu, mpcx_snn_beaver_party_open_vecmul(e.gpu, x, y, t.p, t.p + n, n, ef.p, ef.p + n));
    ef.to_string(e.message_send[peer]);
    e.message_send[2].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
    MSG_CHECK(e.message_recv[peer].size() == 2 * n * W);
    upload_string(efp, e.message_recv[peer]);
    GPU_OK(e.gpu, mpcx_snn_beaver_party_finish_vecmul(e.gpu, (int)e.party_id, t.p, t.p + n, t.p + 2 * n, ef.p, ef.p + n,
                                                      efp.p, efp.p + n, n, z));
}

// shareConvert (four rounds); parties 0/1: a -> y, party 2: no input, no output
void share_convert_dev(SecurennEngine &e, const Dev<int64_t> *a, size_t n, Dev<int64_t> *y) {
    mpcx_ctx *gpu = e.gpu;
    if (e.party_id != 2) {
        const size_t peer = 1 - e.party_id;
        const uint64_t s01 = e.prgs[peer].init_seed;
        Dev<int64_t> x(gpu, n);
        GPU_OK(gpu, mpcx_snn_sc_party_open(gpu, (int)e.party_id, *a, n, 0, n, s01, x));
        e.message_send[peer].clear();
        x.to_string(e.message_send[2]);
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
        e.message_send[2].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2: x bits over Z_67 | delta_j
        MSG_CHECK(e.message_recv[2].size() == n * (ZP + W));
        Dev<uint16_t> xbits(gpu, 64 * n);
        Dev<int64_t> delta(gpu, n), rm1(gpu, n), eta1(gpu, n);
        Dev<int8_t> eta2(gpu, n);
        upload_string(xbits, e.message_recv[2], 0, n * ZP);
        upload_string(delta, e.message_recv[2], n * ZP, n * W);
        GPU_OK(gpu, mpcx_snn_sc_party_pc_inputs(gpu, n, 0, n, s01, eta2, rm1));
        pc_party_dev(e, xbits, eta2, rm1, n); // round 3
        e.message_send[peer].clear();
        e.message_send[2].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4: eta'_j
        MSG_CHECK(e.message_recv[2].size() == n * W);
        upload_string(eta1, e.message_recv[2]);
        GPU_OK(gpu, mpcx_snn_sc_party_finish(gpu, (int)e.party_id, *a, delta, eta1, n, 0, n, s01, *y));
        return;
    }
    const uint64_t sS = e.prgs[2].init_seed;
    e.message_send[0].clear(); // nothing to say in round 1 (parties 0/1 ignore this slot)
    e.message_send[1].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
    MSG_CHECK(e.message_recv[0].size() == n * W && e.message_recv[1].size() == n * W);
    {
        Dev<int64_t> x0(gpu, n), x1(gpu, n), d0(gpu, n), d1(gpu, n);
        Dev<uint16_t> xb0(gpu, 64 * n), xb1(gpu, 64 * n);
        upload_string(x0, e.message_recv[0]);
        upload_string(x1, e.message_recv[1]);
        GPU_OK(gpu, mpcx_snn_sc_helper_split(gpu, x0, x1, n, 0, n, sS, xb0, d0, xb1, d1));
        for (size_t j = 0; j < 2; j++) { // bit shares | delta_j
            e.message_send[j].resize(n * (ZP + W));
            if (n) {
                (j ? xb1 : xb0).down(&e.message_send[j][0], 64 * n);
                (j ? d1 : d0).down(&e.message_send[j][n * ZP], n);
            }
        }
    }
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
    Dev<int8_t> eta1_plain(gpu, n);
    pc_helper_dev(e, n, eta1_plain); // round 3
    Dev<int64_t> e0(gpu, n), e1(gpu, n);
    GPU_OK(gpu, mpcx_snn_sc_helper_eta(gpu, eta1_plain, n, 0, n, sS, e0, e1));
    e0.to_string(e.message_send[0]);
    e1.to_string(e.message_send[1]);
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4
}

// msb over Z_{2^64-1} (= lsb of 2a), result shared over Z_2^64; six rounds
void msb_L1_dev(SecurennEngine &e, const Dev<int64_t> *a, size_t n, Dev<int64_t> *b) {
    mpcx_ctx *gpu = e.gpu;
    if (e.party_id == 2) {
        const uint64_t sS = e.prgs[2].init_seed;
        {
            Dev<int64_t> x0(gpu, n), l0(gpu, n), x1(gpu, n), l1(gpu, n);
            Dev<uint16_t> xb0(gpu, 64 * n), xb1(gpu, 64 * n);
            GPU_OK(gpu, mpcx_snn_msbl1_helper_deal(gpu, n, 0, n, sS, x0, l0, xb0, x1, l1, xb1));
            for (size_t j = 0; j < 2; j++) { // x_j | lsb(x)_j | bits of x over Z_67
                e.message_send[j].resize(n * (2 * W + ZP));
                if (n) {
                    (j ? x1 : x0).down(&e.message_send[j][0], n);
                    (j ? l1 : l0).down(&e.message_send[j][n * W], n);
                    (j ? xb1 : xb0).down(&e.message_send[j][2 * n * W], 64 * n);
                }
            }
        }
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 1
        e.message_send[0].clear();
        e.message_send[1].clear();
        e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2
        Dev<int8_t> beta1_plain(gpu, n);
        pc_helper_dev(e, n, beta1_plain); // round 3
        Dev<int64_t> b0(gpu, n), b1(gpu, n);
        GPU_OK(gpu, mpcx_snn_msbl1_helper_beta(gpu, beta1_plain, n, 0, n, sS, b0, b1));
        b0.to_string(e.message_send[0]);
        b1.to_string(e.message_send[1]);
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
    Dev<int64_t> xj(gpu, n), xlsb(gpu, n), rj(gpu, n), rpeer(gpu, n), R(gpu, n);
    Dev<uint16_t> xbits(gpu, 64 * n);
    upload_string(xj, e.message_recv[2], 0, n * W);
    upload_string(xlsb, e.message_recv[2], n * W, n * W);
    upload_string(xbits, e.message_recv[2], 2 * n * W, n * ZP);
    GPU_OK(gpu, mpcx_snn_msbl1_party_open(gpu, *a, xj, n, rj));
    rj.to_string(e.message_send[peer]);
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 2: open r = 2a + x
    MSG_CHECK(e.message_recv[peer].size() == n * W);
    upload_string(rpeer, e.message_recv[peer]);
    Dev<int8_t> beta0(gpu, n);
    GPU_OK(gpu, mpcx_snn_msbl1_party_pc_inputs(gpu, rj, rpeer, 0, n, s01, beta0, R));
    pc_party_dev(e, xbits, beta0, R, n); // round 3
    e.message_send[2].clear();
    e.message_send[peer].clear();
    e.messenger->send_and_recv(e.message_send, e.message_recv); // round 4: share of beta'
    MSG_CHECK(e.message_recv[2].size() == n * W);
    Dev<int64_t> beta1(gpu, n), gamma(gpu, n), delta(gpu, n), theta(gpu, n);
    upload_string(beta1, e.message_recv[2]);
    GPU_OK(gpu, mpcx_snn_msbl1_party_mid(gpu, (int)e.party_id, beta1, xlsb, beta0, R, n, gamma, delta));
    vecmul_party_dev(e, gamma, delta, n, theta); // rounds 5-6
    GPU_OK(gpu, mpcx_snn_msbl1_party_finish(gpu, gamma, delta, theta, n, *b));
}

} // namespace

// ---- privateCompare: parties 0/1 send masked difference digits to party 2, which learns beta' = beta xor (x > r)
void SecurennEngine::privateCompare(const vector<zpshare> &Xbits_j, const vector<int8_t> &Beta1_plain,
                                    const vector<int64_t> &R_palin, vector<int8_t> &Beta2_plain) {
    const size_t n = Xbits_j.size();
    assert(Beta1_plain.size() == n && R_palin.size() == n);
    Beta2_plain.assign(n, 0);
    if (party_id != 2) {
        Dev<uint16_t> xb(gpu, (const uint16_t *)Xbits_j.data(), 64 * n);
        Dev<int8_t> beta(gpu, Beta1_plain.data(), n);
        Dev<int64_t> r(gpu, R_palin.data(), n);
        pc_party_dev(*this, xb, beta, r, n);
        return;
    }
    Dev<int8_t> b(gpu, n);
    pc_helper_dev(*this, n, b);
    b.down(Beta2_plain.data(), n);
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
    const size_t n = Aj.size();
    Yj.assign(n, 0);
    if (party_id == 2) {
        share_convert_dev(*this, nullptr, n, nullptr);
        return;
    }
    Dev<int64_t> a(gpu, Aj.data(), n), y(gpu, n);
    share_convert_dev(*this, &a, n, &y);
    y.down(Yj.data(), n);
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
    const size_t n = Aj.size();
    Bj.assign(n, 0);
    if (party_id == 2) {
        msb_L1_dev(*this, nullptr, n, nullptr);
        return;
    }
    Dev<int64_t> a(gpu, Aj.data(), n), b(gpu, n);
    msb_L1_dev(*this, &a, n, &b);
    b.down(Bj.data(), n);
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
    const size_t n = Aj.size();
    Bj.assign(n, 0);
    if (party_id == 2) {
        share_convert_dev(*this, nullptr, n, nullptr);
        msb_L1_dev(*this, nullptr, n, nullptr);
        return;
    }
    Dev<int64_t> a(gpu, Aj.data(), n), twoA(gpu, n), y(gpu, n), b(gpu, n);
    GPU_OK(gpu, mpcx_ring_shl1(gpu, a, twoA, n));
    share_convert_dev(*this, &twoA, n, &y);
    msb_L1_dev(*this, &y, n, &b);
    b.down(Bj.data(), n);
}
