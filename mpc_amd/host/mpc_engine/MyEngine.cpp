#include "MyEngine.hpp"

#include "gpu_support.hpp"

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
} // namespace

MyEngine::MyEngine(VirtuleMessenger *_messenger, size_t _party_id) : messenger(_messenger), party_id(_party_id) {
    message_send.resize(3);
    message_recv.resize(3);
    // prgs[k] is the stream shared with party k; prgs[party_id] is the private one (MyEngine.cpp:9-28)
    static const uint64_t pair_key[3][3] = {{KEY0, KEY01, KEY20}, {KEY01, KEY1, KEY12}, {KEY20, KEY12, KEY2}};
    assert(party_id < 3);
    for (size_t k = 0; k < 3; k++) prgs.push_back(PRgenerator(pair_key[party_id][k], 0));
    group = mpcgpu::open_group();
    gpu = group ? mpcx_group_ctx(group, 0) : mpcgpu::open_context();
}

MyEngine::~MyEngine() {
    if (group)
        mpcx_group_shutdown(group);
    else
        mpcx_shutdown(gpu);
}

// ---- additive sharing between parties 0 and 1 out of their common stream; no communication
void MyEngine::share(const VecI64 &X_plain, VecI64 &Xj) {
    const size_t n = X_plain.size();
    Xj.resize(n);
    for (size_t i = 0; i < n; i++) Xj[i] = 0;
    if (party_id == 2 || n == 0) return;
    PRgenerator &g = prgs[1 - party_id];
    Dev<int64_t> x(gpu, X_plain.data(), n), out(gpu, n);
    GPU_OK(gpu, mpcx_snn_share_local(gpu, (int)party_id, x, n, g.init_seed, g.pops, out));
    g.advance(gpu, n);
    out.down(Xj.data(), n);
}

void MyEngine::share_matrix(const MatrixXl &X_plain, MatrixXl &Xj) {
    const size_t rows = X_plain.rows(), cols = X_plain.cols(), n = rows * cols;
    Xj.resize(rows, cols);
    for (size_t i = 0; i < n; i++) Xj.data()[i] = 0;
    if (party_id == 2 || n == 0) return;
    PRgenerator &g = prgs[1 - party_id];
    // the words are popped in row-major element order (:81-87); Eigen stores column-major
    Dev<int64_t> xc(gpu, X_plain.data(), n), xr(gpu, n), out(gpu, n), outc(gpu, n);
    GPU_OK(gpu, mpcx_relayout(gpu, xc, xr, rows, cols, 1, 0));
    GPU_OK(gpu, mpcx_snn_share_local(gpu, (int)party_id, xr, n, g.init_seed, g.pops, out));
    g.advance(gpu, n);
    GPU_OK(gpu, mpcx_relayout(gpu, out, outc, rows, cols, 0, 1));
    outc.down(Xj.data(), n);
}

// ---- openings: parties 0 and 1 swap shares and copy them to party 2
void MyEngine::reveal(const VecI64 &Xj, VecI64 &X_plain) {
    const size_t n = Xj.size();
    X_plain.resize(n);
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vecI642string(Xj, message_send[1 - party_id]);
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
        upload_string(u, message_recv[1 - party_id]);
        v.up(Xj.data(), n);
    }
    GPU_OK(gpu, mpcx_ring_add(gpu, u, v, out, n));
    out.down(X_plain.data(), n);
}

void MyEngine::reveal_matrix(const MatrixXl &Xj, MatrixXl &X_plain) {
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

void MyEngine::reveal_vec_bools(const vecbool &Xj, vecbool &X_plain) {
    const size_t n = Xj.size();
    X_plain.resize(n);
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vecbool2string(Xj, message_send[1 - party_id]);
        message_send[2] = message_send[1 - party_id];
    }
    messenger->send_and_recv(message_send, message_recv);
    Dev<int8_t> u(gpu, n), v(gpu, n), out(gpu, n);
    if (party_id == 2) {
        MSG_CHECK(message_recv[0].size() == n && message_recv[1].size() == n);
        upload_string(u, message_recv[0]);
        upload_string(v, message_recv[1]);
    } else {
        MSG_CHECK(message_recv[1 - party_id].size() == n);
        upload_string(u, message_recv[1 - party_id]);
        v.up(Xj.data(), n);
    }
    GPU_OK(gpu, mpcx_xor_bools(gpu, u, v, out, n));
    out.down(X_plain.data(), n);
}

void MyEngine::reveal_L_1(const VecI64 &Xj, VecI64 &X_plain) {
    const size_t n = Xj.size();
    X_plain.resize(n);
    if (party_id == 2) {
        message_send[0].clear();
        message_send[1].clear();
    } else {
        vecI642string(Xj, message_send[1 - party_id]);
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
        upload_string(u, message_recv[1 - party_id]);
        v.up(Xj.data(), n);
    }
    GPU_OK(gpu, mpcx_snn_add_L1(gpu, v, u, out, n));
    out.down(X_plain.data(), n);
}

void MyEngine::reveal_zp(const vector<zpshare> &Xbits_j, vector<zpshare> &Xbits_plain) {
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

// ---- Beaver products, one round: party 2 sends C_j, parties 0/1 swap E_j | F_j in the same round
void MyEngine::matrix_mul(const MatrixXl &Xj, const MatrixXl &Yj, MatrixXl &Zj) {
#if defined MYDEBUG
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
    Zj.resize(rows, cols);
    const bool spread = group && rows && lens && cols; // row blocks over the group's GPUs
    if (party_id == 2) {
        if (spread) {
            message_send[0].resize(nc * W);
            message_send[1].resize(nc * W);
            if (mpcx_group_my_beaver_dealer_matmul(group, prgs[0].init_seed, prgs[1].init_seed, prgs[2].init_seed, rows, lens,
                                                   cols, (int64_t *)&message_send[0][0], (int64_t *)&message_send[1][0]) != 0) {
                fprintf(stderr, "mpcx_group_my_beaver_dealer_matmul: %s\n", mpcx_group_last_error(group));
                abort();
            }
        } else {
            Dev<int64_t> c0(gpu, nc), c1(gpu, nc);
            GPU_OK(gpu, mpcx_my_beaver_dealer_matmul(gpu, prgs[0].init_seed, prgs[1].init_seed, prgs[2].init_seed, rows,
                                                     lens, cols, c0, c1));
            c0.to_string(message_send[0]);
            c1.to_string(message_send[1]);
        }
        messenger->send_and_recv(message_send, message_recv);
        return;
    }
    const size_t peer = 1 - party_id;
    const uint64_t s2 = prgs[2].init_seed;
    Dev<int64_t> ef(gpu, na + nb), ef_peer(gpu, na + nb), c(gpu, nc), z(gpu, nc);
    {
        Dev<int64_t> x(gpu, Xj.data(), na), y(gpu, Yj.data(), nb);
        GPU_OK(gpu, mpcx_my_beaver_party_open_matmul(gpu, x, y, 1, s2, rows, lens, cols, ef));
    }
    ef.to_string(message_send[peer]);
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[2].size() == nc * W && message_recv[peer].size() == (na + nb) * W);
    if (spread) { // straight from the three messages; the product comes back row-major
        VecI64 zr(nc);
        if (mpcx_group_my_beaver_party_finish_matmul(group, (int)party_id, s2, (const int64_t *)message_recv[2].data(),
                                                     (const int64_t *)message_send[peer].data(),
                                                     (const int64_t *)message_recv[peer].data(), rows, lens, cols,
                                                     zr.data()) != 0) {
            fprintf(stderr, "mpcx_group_my_beaver_party_finish_matmul: %s\n", mpcx_group_last_error(group));
            abort();
        }
        vecI642matrix(zr, Zj);
        return;
    }
    upload_string(c, message_recv[2]);
    upload_string(ef_peer, message_recv[peer]);
    GPU_OK(gpu, mpcx_my_beaver_party_finish_matmul(gpu, (int)party_id, s2, c, ef, ef_peer, rows, lens, cols, 1, z));
    z.down(Zj.data(), nc);
}

void MyEngine::vector_mul(const VecI64 &Xj, const VecI64 &Yj, VecI64 &Zj) {
#if defined MYDEBUG
    mpcgpu::AtExit check([&] {
        VecI64 xd, yd, zd;
        reveal(Xj, xd);
        reveal(Yj, yd);
        reveal(Zj, zd);
        for (size_t i = 0; i < (size_t)zd.size(); i++) assert((uint64_t)zd[i] == (uint64_t)xd[i] * (uint64_t)yd[i]);
    });
#endif
    // Like the protocols below, written over element ranges (mpcgpu::run): the engine's GPU takes the whole vector, the
    // members of an MPCX_DEVICES group one contiguous range each, with their own slices of every message section.
    const size_t n = Xj.size(), G = members(*this);
    assert((size_t)Yj.size() == n);
    Zj.resize(n);
    if (party_id == 2) {
        for (size_t j = 0; j < 2; j++) message_send[j].resize(n * W);
        run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> c0(ctx, cnt), c1(ctx, cnt);
            GPU_OK(ctx, mpcx_my_beaver_dealer_vecmul(ctx, prgs[0].init_seed, prgs[1].init_seed, prgs[2].init_seed, n, i0, cnt,
                                                     c0, c1));
            put(c0, message_send[0], 0, i0, cnt);
            put(c1, message_send[1], 0, i0, cnt);
        });
        messenger->send_and_recv(message_send, message_recv);
        return;
    }
    const size_t peer = 1 - party_id;
    const uint64_t s2 = prgs[2].init_seed;
    Bufs<int64_t> ef(G);
    message_send[peer].resize(2 * n * W); // E | F
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> x(ctx, Xj.data() + i0, cnt), y(ctx, Yj.data() + i0, cnt);
        Dev<int64_t> &efm = fresh(ef[m], ctx, 2 * cnt);
        GPU_OK(ctx, mpcx_my_beaver_party_open_vecmul(ctx, x, y, s2, n, i0, cnt, efm));
        put_sections(efm, message_send[peer], 0, 2, n, i0, cnt);
    });
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[2].size() == n * W && message_recv[peer].size() == 2 * n * W);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> c(ctx, cnt), efp(ctx, 2 * cnt), z(ctx, cnt);
        get(c, message_recv[2], 0, i0, cnt);
        get_sections(efp, message_recv[peer], 0, 2, n, i0, cnt);
        GPU_OK(ctx, mpcx_my_beaver_party_finish_vecmul(ctx, (int)party_id, s2, c, ef[m]->p, efp, n, i0, cnt, z));
        if (cnt) z.down(Zj.data() + i0, cnt);
    });
}

// ---- privateCompare: parties 0/1 send masked difference digits to party 2, which learns beta' = beta xor (x > r)
void MyEngine::privateCompare(const vector<zpshare> &Xbits_j, const vecbool &Beta1_plain, const VecI64 &R_palin,
                              vecbool &Beta2_plain) {
    const size_t n = Xbits_j.size();
    assert(Beta1_plain.size() == n && (size_t)R_palin.size() == n);
    Beta2_plain.resize(n);
    if (party_id != 2) {
        message_send[1 - party_id].clear();
        message_send[2].resize(n * ZP);
        run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<uint16_t> xb(ctx, (const uint16_t *)Xbits_j.data() + 64 * i0, 64 * cnt), d(ctx, 64 * cnt);
            Dev<int8_t> beta(ctx, Beta1_plain.data() + i0, cnt);
            Dev<int64_t> r(ctx, R_palin.data() + i0, cnt);
            GPU_OK(ctx, mpcx_my_pc_party(ctx, (int)party_id, xb, beta, r, i0, cnt, prgs[1 - party_id].init_seed, d));
            put(d, message_send[2], 0, i0, cnt, 64);
        });
        messenger->send_and_recv(message_send, message_recv);
        return;
    }
    message_send[0].clear();
    message_send[1].clear();
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[0].size() == n * ZP && message_recv[1].size() == n * ZP);
    run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
        Dev<uint16_t> d0(ctx, 64 * cnt), d1(ctx, 64 * cnt);
        Dev<int8_t> b(ctx, cnt);
        get(d0, message_recv[0], 0, i0, cnt, 64);
        get(d1, message_recv[1], 0, i0, cnt, 64);
        GPU_OK(ctx, mpcx_snn_pc_helper(ctx, d0, d1, cnt, b));
        if (cnt) b.down(Beta2_plain.data() + i0, cnt);
    });
}

// ---- msb of a in [-2^62, 2^62): three rounds, XOR-shared result
void MyEngine::msb(const VecI64 &Aj, vecbool &Bj) {
#if defined MYDEBUG // MyEngine.cpp:582: XOR shares of the sign bit
    mpcgpu::AtExit check([&] {
        VecI64 ad;
        vecbool bd;
        reveal(Aj, ad);
        reveal_vec_bools(Bj, bd);
        for (size_t i = 0; i < bd.size(); i++) assert(bd[i] == (int8_t)(ad[i] < 0));
    });
#endif
    const size_t n = Aj.size(), G = members(*this);
    Bj.assign(n, 0);
    if (party_id == 2) {
        const uint64_t s0 = prgs[0].init_seed, s1 = prgs[1].init_seed, sS = prgs[2].init_seed;
        for (size_t j = 0; j < 2; j++) message_send[j].resize(n + n * ZP); // lsb share | bit shares (:502-508)
        run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<int8_t> l0(ctx, cnt), l1(ctx, cnt);
            Dev<uint16_t> xb0(ctx, 64 * cnt), xb1(ctx, 64 * cnt);
            GPU_OK(ctx, mpcx_my_msb_helper_deal(ctx, s0, s1, sS, i0, cnt, l0, xb0, l1, xb1));
            put(l0, message_send[0], 0, i0, cnt);
            put(xb0, message_send[0], n, i0, cnt, 64);
            put(l1, message_send[1], 0, i0, cnt);
            put(xb1, message_send[1], n, i0, cnt, 64);
        });
        messenger->send_and_recv(message_send, message_recv); // round 1
        message_send[0].clear();
        message_send[1].clear();
        messenger->send_and_recv(message_send, message_recv); // round 2: privateCompare as the helper
        MSG_CHECK(message_recv[0].size() == n * ZP && message_recv[1].size() == n * ZP);
        for (size_t j = 0; j < 2; j++) message_send[j].resize(n);
        run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<uint16_t> d0(ctx, 64 * cnt), d1(ctx, 64 * cnt);
            Dev<int8_t> beta1(ctx, cnt), b0(ctx, cnt), b1(ctx, cnt);
            get(d0, message_recv[0], 0, i0, cnt, 64);
            get(d1, message_recv[1], 0, i0, cnt, 64);
            GPU_OK(ctx, mpcx_snn_pc_helper(ctx, d0, d1, cnt, beta1));
            GPU_OK(ctx, mpcx_my_msb_helper_beta(ctx, beta1, n, i0, cnt, sS, b0, b1));
            put(b0, message_send[0], 0, i0, cnt);
            put(b1, message_send[1], 0, i0, cnt);
        });
        messenger->send_and_recv(message_send, message_recv); // round 3
        return;
    }
    const size_t peer = 1 - party_id;
    const uint64_t s01 = prgs[peer].init_seed, s2 = prgs[2].init_seed;
    Bufs<int64_t> rj(G), R(G);
    Bufs<int8_t> beta0(G), xlsb(G);
    message_send[2].clear();
    message_send[peer].resize(n * W);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> a(ctx, Aj.data() + i0, cnt);
        GPU_OK(ctx, mpcx_my_msb_party_open(ctx, a, s2, i0, cnt, fresh(rj[m], ctx, cnt)));
        put(*rj[m], message_send[peer], 0, i0, cnt);
    });
    messenger->send_and_recv(message_send, message_recv); // round 1
    MSG_CHECK(message_recv[peer].size() == n * W && message_recv[2].size() == n + n * ZP);
    message_send[peer].clear();
    message_send[2].resize(n * ZP);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> rpeer(ctx, cnt);
        Dev<uint16_t> xbits(ctx, 64 * cnt), d(ctx, 64 * cnt);
        get(rpeer, message_recv[peer], 0, i0, cnt);
        get(fresh(xlsb[m], ctx, cnt), message_recv[2], 0, i0, cnt);
        get(xbits, message_recv[2], n, i0, cnt, 64);
        GPU_OK(ctx, mpcx_my_msb_party_pc_inputs(ctx, rj[m]->p, rpeer, i0, cnt, s01, fresh(beta0[m], ctx, cnt),
                                                fresh(R[m], ctx, cnt)));
        GPU_OK(ctx, mpcx_my_pc_party(ctx, (int)party_id, xbits, beta0[m]->p, R[m]->p, i0, cnt, s01, d));
        put(d, message_send[2], 0, i0, cnt, 64);
    });
    messenger->send_and_recv(message_send, message_recv); // round 2
    message_send[2].clear();
    message_send[peer].clear();
    messenger->send_and_recv(message_send, message_recv); // round 3: share of beta'
    MSG_CHECK(message_recv[2].size() == n);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int8_t> beta1(ctx, cnt), out(ctx, cnt);
        get(beta1, message_recv[2], 0, i0, cnt);
        GPU_OK(ctx, mpcx_my_msb_party_finish(ctx, (int)party_id, beta1, beta0[m]->p, xlsb[m]->p, R[m]->p, cnt, out));
        if (cnt) out.down(Bj.data() + i0, cnt);
    });
}

// ---- x * a for an XOR-shared boolean a, one round
void MyEngine::bit_inject(const VecI64 &Xj, const vecbool &Aj, VecI64 &Zj) {
#if defined MYDEBUG // :752: Z = X * a
    mpcgpu::AtExit check([&] {
        VecI64 xd, zd;
        vecbool ad;
        reveal(Xj, xd);
        reveal_vec_bools(Aj, ad);
        reveal(Zj, zd);
        for (size_t i = 0; i < (size_t)zd.size(); i++) assert(zd[i] == (ad[i] ? xd[i] : 0));
    });
#endif
    const size_t n = Xj.size(), G = members(*this);
    assert(Aj.size() == n);
    Zj.resize(n);
    if (party_id == 2) {
        for (size_t j = 0; j < 2; j++) message_send[j].resize(4 * n * W); // V | W | S | T
        run(*this, n, [&](mpcx_ctx *ctx, int, uint64_t i0, uint64_t cnt) {
            Dev<int64_t> m0(ctx, 4 * cnt), m1(ctx, 4 * cnt);
            GPU_OK(ctx, mpcx_my_bit_inject_dealer(ctx, prgs[0].init_seed, prgs[1].init_seed, prgs[2].init_seed, n, i0, cnt, m0.p,
                                                  m0.p + cnt, m0.p + 2 * cnt, m0.p + 3 * cnt, m1.p, m1.p + cnt,
                                                  m1.p + 2 * cnt, m1.p + 3 * cnt));
            put_sections(m0, message_send[0], 0, 4, n, i0, cnt);
            put_sections(m1, message_send[1], 0, 4, n, i0, cnt);
        });
        messenger->send_and_recv(message_send, message_recv);
        return;
    }
    const size_t peer = 1 - party_id;
    const uint64_t s2 = prgs[2].init_seed;
    Bufs<int64_t> y(G);
    Bufs<int8_t> c(G);
    message_send[peer].resize(n * W + n); // y | c (:621-623)
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> x(ctx, Xj.data() + i0, cnt);
        Dev<int8_t> a(ctx, Aj.data() + i0, cnt);
        GPU_OK(ctx, mpcx_my_bit_inject_party_pre(ctx, x, a, s2, i0, cnt, fresh(y[m], ctx, cnt), fresh(c[m], ctx, cnt)));
        put(*y[m], message_send[peer], 0, i0, cnt);
        put(*c[m], message_send[peer], n * W, i0, cnt);
    });
    message_send[2].clear();
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[peer].size() == n * W + n && message_recv[2].size() == 4 * n * W);
    run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        Dev<int64_t> ypeer(ctx, cnt), deal(ctx, 4 * cnt), z(ctx, cnt);
        Dev<int8_t> cpeer(ctx, cnt);
        get(ypeer, message_recv[peer], 0, i0, cnt);
        get(cpeer, message_recv[peer], n * W, i0, cnt);
        get_sections(deal, message_recv[2], 0, 4, n, i0, cnt);
        GPU_OK(ctx, mpcx_my_bit_inject_party_post(ctx, (int)party_id, s2, i0, cnt, y[m]->p, c[m]->p, ypeer, cpeer, deal.p,
                                                  deal.p + cnt, deal.p + 2 * cnt, deal.p + 3 * cnt, z));
        if (cnt) z.down(Zj.data() + i0, cnt);
    });
}
