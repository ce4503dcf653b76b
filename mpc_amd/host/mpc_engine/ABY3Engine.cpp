#include "ABY3Engine.hpp"

#include <memory>

#include "gpu_support.hpp"

using mpcgpu::Dev;
using mpcgpu::upload_string;

namespace {
struct Peers {
    size_t next, prev;
    explicit Peers(size_t id) : next((id + 1) % 3), prev((id + 2) % 3) {}
};
} // namespace

ABY3Engine::ABY3Engine(VirtuleMessenger *_messenger, size_t _party_id) : messenger(_messenger), party_id(_party_id) {
    message_send.resize(3);
    message_recv.resize(3);
    // prgs[k] is the stream shared with party k; prgs[party_id] is the private one
    static const uint64_t pair_key[3][3] = {{KEY0, KEY01, KEY20}, {KEY01, KEY1, KEY12}, {KEY20, KEY12, KEY2}};
    assert(party_id < 3);
    for (size_t k = 0; k < 3; k++) prgs.push_back(PRgenerator(pair_key[party_id][k], 0));
    group = mpcgpu::open_group();
    gpu = group ? mpcx_group_ctx(group, 0) : mpcgpu::open_context();
}

// a share pair kept on the device, column-major as Eigen stores it -- or, for an engine that spreads its products over a
// group of GPUs, a row-major copy on every member (mpcx_group_operand: the replicated second operand of matrix_mul)
struct ABY3Engine::Pinned {
    const MatPair *key;
    size_t rows, cols;
    std::unique_ptr<Dev<int64_t>> s0, s1;
    mpcx_group *group = nullptr;
    mpcx_group_operand *gop = nullptr;
    mpcx_ctx *ctx = nullptr;
    // the pair as a product's second operand (residue planes where the library uses them), made at its first use as one
    mpcx_aby3_operand *as_second = nullptr;
    Pinned(mpcx_ctx *c, mpcx_group *g, const MatPair &M) : key(&M), rows(M[0].rows()), cols(M[0].cols()), group(g), ctx(c) {
        if (group) {
            if (mpcx_group_operand_upload(group, M[0].data(), M[1].data(), rows, cols, 1, &gop) != 0) {
                fprintf(stderr, "mpcx_group_operand_upload: %s\n", mpcx_group_last_error(group));
                abort();
            }
        } else {
            s0.reset(new Dev<int64_t>(ctx, M[0].data(), rows * cols));
            s1.reset(new Dev<int64_t>(ctx, M[1].data(), rows * cols));
        }
    }
    ~Pinned() {
        if (as_second) mpcx_aby3_operand_free(ctx, as_second);
        if (gop) mpcx_group_operand_free(group, gop);
    }
};
ABY3Engine::Pinned *ABY3Engine::find_pinned(const MatPair &M) const {
    for (Pinned *p : pinned)
        if (p->key == &M) return p;
    return nullptr;
}
void ABY3Engine::pin(const MatPair &M) {
    assert(M[0].rows() == M[1].rows() && M[0].cols() == M[1].cols());
    unpin(M);
    if (M[0].rows() && M[0].cols()) pinned.push_back(new Pinned(gpu, group, M));
}
void ABY3Engine::unpin(const MatPair &M) {
    for (size_t i = 0; i < pinned.size(); i++)
        if (pinned[i]->key == &M) {
            delete pinned[i];
            pinned.erase(pinned.begin() + i);
            return;
        }
}

ABY3Engine::~ABY3Engine() {
    for (Pinned *p : pinned) delete p;
    pinned.clear();
    if (group)
        mpcx_group_shutdown(group);
    else
        mpcx_shutdown(gpu);
}

// ---- the same protocols over a group of GPUs.  Every member works on a contiguous range of the elements (whole
// multiples of 64, so that the dense bit strings of msb split at word boundaries) with its own device buffers, and
// moves its slices of the operands, of the messages and of the results itself; the rounds and the wire bytes are
// those of the single-GPU methods.
namespace {
template <class T> using DevVec = std::vector<std::unique_ptr<Dev<T>>>;
constexpr uint64_t RANGE_GRANULE = 64;
constexpr size_t W8 = sizeof(int64_t);

void add_prepare_group(ABY3Engine &e, const array<VecI64, 2> &Xj, array<VecI64, 2> &Gj, array<VecI64, 2> &Pj) {
    const Peers pe(e.party_id);
    const size_t n = Xj[0].size(), G = mpcx_group_size(e.group);
    DevVec<int64_t> x0(G), x1(G), c0(G), c1(G), p0(G), p1(G), g0(G);
    string &out = e.message_send[pe.prev];
    e.message_send[pe.next].clear();
    out.resize(n * W8);
    mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        x0[m].reset(new Dev<int64_t>(ctx, Xj[0].data() + i0, cnt));
        x1[m].reset(new Dev<int64_t>(ctx, Xj[1].data() + i0, cnt));
        c0[m].reset(new Dev<int64_t>(ctx, cnt));
        GPU_OK(ctx, mpcx_aby3_add_prepare_step1(ctx, *x0[m], *x1[m], cnt, *c0[m]));
        if (cnt) c0[m]->down(&out[i0 * W8], cnt);
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv);
    const string &in = e.message_recv[pe.next];
    MSG_CHECK(in.size() == n * W8);
    Pj[0].resize(n);
    Pj[1].resize(n);
    mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        c1[m].reset(new Dev<int64_t>(ctx, (const int64_t *)in.data() + i0, cnt));
        p0[m].reset(new Dev<int64_t>(ctx, cnt));
        p1[m].reset(new Dev<int64_t>(ctx, cnt));
        g0[m].reset(new Dev<int64_t>(ctx, cnt));
        GPU_OK(ctx, mpcx_aby3_add_prepare_step2(ctx, *x0[m], *x1[m], *c0[m], *c1[m], cnt, *p0[m], *p1[m], *g0[m]));
        if (cnt) {
            g0[m]->down(&out[i0 * W8], cnt);
            p0[m]->down(Pj[0].data() + i0, cnt);
            p1[m]->down(Pj[1].data() + i0, cnt);
        }
    });
    e.messenger->send_and_recv(e.message_send, e.message_recv);
    string2vecI64(e.message_send[pe.prev], Gj[0]);
    string2vecI64(e.message_recv[pe.next], Gj[1]);
}

// The messages of the carry tree are booleans, one byte each on the wire (vecbool2string).  Between device and host
// they travel as bit strings -- an eighth of the bytes over the link -- and the bytes are made and read by the host's
// copy threads (mpcx_d2h_bits / mpcx_h2d_bits); MPCX_MSB_WIRE=bytes moves the bytes themselves, for comparison.  The
// device buffers (64 bytes per element) hold either form.
bool msb_as_bits() {
    static const bool on = [] {
        const char *e = getenv("MPCX_MSB_WIRE");
        return !(e && strcmp(e, "bytes") == 0);
    }();
    return on;
}
void msb_first(mpcx_ctx *ctx, uint64_t *state, size_t cnt, int8_t *msg_out) {
    if (msb_as_bits())
        GPU_OK(ctx, mpcx_aby3_msb_round_pre_bits(ctx, state, cnt, 32, (uint32_t *)msg_out));
    else
        GPU_OK(ctx, mpcx_aby3_msb_round_pre(ctx, state, cnt, 32, msg_out));
}
void msb_next(mpcx_ctx *ctx, uint64_t *state, size_t cnt, size_t temp, const int8_t *msg_in, int8_t *msg_out) {
    if (msb_as_bits())
        GPU_OK(ctx, mpcx_aby3_msb_round_mid_bits(ctx, state, cnt, temp, (const uint32_t *)msg_in, (uint32_t *)msg_out));
    else
        GPU_OK(ctx, mpcx_aby3_msb_round_mid(ctx, state, cnt, temp, msg_in, msg_out));
}
void msb_end(mpcx_ctx *ctx, const uint64_t *state, size_t cnt, const int8_t *msg_in, int8_t *z0, int8_t *z1) {
    if (msb_as_bits())
        GPU_OK(ctx, mpcx_aby3_msb_last_bits(ctx, state, cnt, (const uint32_t *)msg_in, z0, z1));
    else
        GPU_OK(ctx, mpcx_aby3_msb_last(ctx, state, cnt, msg_in, z0, z1));
}
void msb_msg_down(mpcx_ctx *ctx, char *host, const int8_t *dev, size_t booleans) {
    if (!booleans) return;
    GPU_OK(ctx, msb_as_bits() ? mpcx_d2h_bits(ctx, host, dev, booleans) : mpcx_d2h(ctx, host, dev, booleans));
}
void msb_msg_up(mpcx_ctx *ctx, int8_t *dev, const char *host, size_t booleans) {
    if (!booleans) return;
    GPU_OK(ctx, msb_as_bits() ? mpcx_h2d_bits(ctx, dev, host, booleans) : mpcx_h2d(ctx, dev, host, booleans));
}

void msb_group(ABY3Engine &e, const array<VecI64, 2> &Gj, const array<VecI64, 2> &Pj, array<vecbool, 2> &Zj) {
    const Peers pe(e.party_id);
    const size_t n = Gj[0].size(), G = mpcx_group_size(e.group);
    DevVec<int64_t> g0(G), g1(G), p0(G), p1(G);
    DevVec<uint64_t> state(G);
    DevVec<int8_t> msg_in(G), msg_out(G), z0(G), z1(G);
    string &out = e.message_send[pe.prev];
    e.message_send[pe.next].clear();
    out.resize(64 * n);
    mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        g0[m].reset(new Dev<int64_t>(ctx, Gj[0].data() + i0, cnt));
        g1[m].reset(new Dev<int64_t>(ctx, Gj[1].data() + i0, cnt));
        p0[m].reset(new Dev<int64_t>(ctx, Pj[0].data() + i0, cnt));
        p1[m].reset(new Dev<int64_t>(ctx, Pj[1].data() + i0, cnt));
        state[m].reset(new Dev<uint64_t>(ctx, 6 * cnt));
        msg_in[m].reset(new Dev<int8_t>(ctx, 64 * cnt));
        msg_out[m].reset(new Dev<int8_t>(ctx, 64 * cnt));
        z0[m].reset(new Dev<int8_t>(ctx, cnt));
        z1[m].reset(new Dev<int8_t>(ctx, cnt));
        GPU_OK(ctx, mpcx_aby3_msb_begin(ctx, *g0[m], *g1[m], *p0[m], *p1[m], cnt, *state[m]));
        msb_first(ctx, *state[m], cnt, *msg_out[m]);
        msb_msg_down(ctx, &out[64 * i0], *msg_out[m], 64 * cnt);
    });
    Zj[0].resize(n);
    Zj[1].resize(n);
    for (size_t temp = 32; temp != 0; temp /= 2) {
        const size_t per = 2 * temp; // message bytes per element of this round
        out.resize(per * n);
        e.messenger->send_and_recv(e.message_send, e.message_recv);
        const string &in = e.message_recv[pe.next];
        MSG_CHECK(in.size() == per * n);
        if (temp != 1) out.resize(per / 2 * n); // the next message is written in place, slice by slice
        mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            if (!cnt) return;
            msb_msg_up(ctx, *msg_in[m], in.data() + per * i0, per * cnt);
            if (temp != 1) {
                msb_next(ctx, *state[m], cnt, temp, *msg_in[m], *msg_out[m]);
                msb_msg_down(ctx, &out[per / 2 * i0], *msg_out[m], per / 2 * cnt);
            } else {
                msb_end(ctx, *state[m], cnt, *msg_in[m], *z0[m], *z1[m]);
                z0[m]->down(Zj[0].data() + i0, cnt);
                z1[m]->down(Zj[1].data() + i0, cnt);
            }
        });
    }
}

// the sections of a bit_inject message hold n words each; a member's call produces / consumes `cnt` words per section
void sections_down(const Dev<int64_t> &d, string &s, size_t sections, size_t n, size_t i0, size_t cnt) {
    for (size_t k = 0; k < sections && cnt; k++) d.down(&s[(k * n + i0) * W8], cnt, k * cnt);
}
void sections_up(mpcx_ctx *ctx, Dev<int64_t> &d, const string &s, size_t sections, size_t n, size_t i0, size_t cnt) {
    for (size_t k = 0; k < sections && cnt; k++) GPU_OK(ctx, mpcx_h2d(ctx, d.p + k * cnt, s.data() + (k * n + i0) * W8, cnt * W8));
}

void bit_inject_group(ABY3Engine &e, const array<VecI64, 2> &Aj, const array<vecbool, 2> &Bj, array<VecI64, 2> &Cj) {
    const size_t n = Aj[0].size(), G = mpcx_group_size(e.group), id = e.party_id;
    // party p talks to a = (p + 1) % 3 and b = (p + 2) % 3; message sizes in sections of n words (:590-620, :709-790, :875-939)
    const size_t a = (id + 1) % 3, b = (id + 2) % 3;
    static const size_t send_sections[3][3] = {{0, 2, 3}, {2, 0, 3}, {1, 1, 0}}; // [from][to]
    // the two streams of this party, in the order the C ABI takes them
    PRgenerator &sA = e.prgs[id == 0 ? 1 : 0], &sB = e.prgs[id == 2 ? 1 : 2]; // P0: s01, s20; P1: s01, s12; P2: s20, s12
    const uint64_t popA = sA.pops, popB = sB.pops;
    DevVec<int64_t> a0(G), a1(G), toA(G), toB(G), fromA(G), fromB(G), c0(G), c1(G);
    DevVec<int8_t> b0(G), b1(G);
    for (auto &m : e.message_send) m.clear();
    e.message_send[a].resize(send_sections[id][a] * n * W8);
    e.message_send[b].resize(send_sections[id][b] * n * W8);
    mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        a0[m].reset(new Dev<int64_t>(ctx, Aj[0].data() + i0, cnt));
        a1[m].reset(new Dev<int64_t>(ctx, Aj[1].data() + i0, cnt));
        b0[m].reset(new Dev<int8_t>(ctx, Bj[0].data() + i0, cnt));
        b1[m].reset(new Dev<int8_t>(ctx, Bj[1].data() + i0, cnt));
        toA[m].reset(new Dev<int64_t>(ctx, send_sections[id][a] * cnt));
        toB[m].reset(new Dev<int64_t>(ctx, send_sections[id][b] * cnt));
        if (id == 0) // to party 1, to party 2
            GPU_OK(ctx, mpcx_aby3_bit_inject_p0_pre(ctx, *a0[m], *b0[m], *b1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toA[m], *toB[m]));
        else if (id == 1) // to party 0 (= b), to party 2 (= a)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p1_pre(ctx, *a0[m], *a1[m], *b0[m], *b1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toB[m], *toA[m]));
        else // to party 0 (= a), to party 1 (= b)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p2_pre(ctx, *b0[m], *b1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toA[m], *toB[m]));
        sections_down(*toA[m], e.message_send[a], send_sections[id][a], n, i0, cnt);
        sections_down(*toB[m], e.message_send[b], send_sections[id][b], n, i0, cnt);
    });
    sA.advance(e.gpu, (id == 2 ? 3 : 6) * n);
    sB.advance(e.gpu, 3 * n);
    e.messenger->send_and_recv(e.message_send, e.message_recv);
    MSG_CHECK(e.message_recv[a].size() == send_sections[a][id] * n * W8 && e.message_recv[b].size() == send_sections[b][id] * n * W8);
    Cj[0].resize(n);
    Cj[1].resize(n);
    mpcgpu::group_for_each(e.group, n, RANGE_GRANULE, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        fromA[m].reset(new Dev<int64_t>(ctx, send_sections[a][id] * cnt));
        fromB[m].reset(new Dev<int64_t>(ctx, send_sections[b][id] * cnt));
        c0[m].reset(new Dev<int64_t>(ctx, cnt));
        c1[m].reset(new Dev<int64_t>(ctx, cnt));
        sections_up(ctx, *fromA[m], e.message_recv[a], send_sections[a][id], n, i0, cnt);
        sections_up(ctx, *fromB[m], e.message_recv[b], send_sections[b][id], n, i0, cnt);
        if (id == 0) // from party 1, from party 2
            GPU_OK(ctx, mpcx_aby3_bit_inject_p0_post(ctx, *b0[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromA[m], *fromB[m], *c0[m], *c1[m]));
        else if (id == 1) // from party 0 (= b), from party 2 (= a)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p1_post(ctx, *b1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromB[m], *fromA[m], *c0[m], *c1[m]));
        else // from party 0 (= a), from party 1 (= b)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p2_post(ctx, *b0[m], *b1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromA[m], *fromB[m], *c0[m], *c1[m]));
        if (cnt) {
            c0[m]->down(Cj[0].data() + i0, cnt);
            c1[m]->down(Cj[1].data() + i0, cnt);
        }
    });
}
} // namespace

// ---- sharing: R = pop(next) - pop(prev) (+ X on party 0); send share 0 to prev, share 1 comes from next
void ABY3Engine::share(const VecI64 &X_plain, array<VecI64, 2> &Xj) {
    const size_t n = X_plain.size();
    const Peers pe(party_id);
    Dev<int64_t> out(gpu, n);
    {
        Dev<int64_t> x(gpu, X_plain.data(), n);
        GPU_OK(gpu, mpcx_aby3_share_local(gpu, party_id == 0 ? x.p : nullptr, n, prgs[pe.next].init_seed,
                                          prgs[pe.next].pops, prgs[pe.prev].init_seed, prgs[pe.prev].pops, out));
    }
    prgs[pe.next].advance(gpu, n);
    prgs[pe.prev].advance(gpu, n);
    message_send[pe.next].clear();
    out.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecI64(message_send[pe.prev], Xj[0]);
    string2vecI64(message_recv[pe.next], Xj[1]);
}

void ABY3Engine::share_matrix(const MatrixXl &X_plain, array<MatrixXl, 2> &Xj) {
    const size_t rows = X_plain.rows(), cols = X_plain.cols(), n = rows * cols;
    const Peers pe(party_id);
    Dev<int64_t> out(gpu, n);
    {
        Dev<int64_t> xc(gpu, X_plain.data(), n), xr(gpu, n);
        if (party_id == 0 && n) GPU_OK(gpu, mpcx_relayout(gpu, xc, xr, rows, cols, 1, 0)); // the mask order is row-major
        GPU_OK(gpu, mpcx_aby3_share_local(gpu, party_id == 0 ? xr.p : nullptr, n, prgs[pe.next].init_seed,
                                          prgs[pe.next].pops, prgs[pe.prev].init_seed, prgs[pe.prev].pops, out));
    }
    prgs[pe.next].advance(gpu, n);
    prgs[pe.prev].advance(gpu, n);
    message_send[pe.next].clear();
    out.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    Xj[0].resize(rows, cols);
    Xj[1].resize(rows, cols);
    string2matrix(message_send[pe.prev], Xj[0]);
    string2matrix(message_recv[pe.next], Xj[1]);
}

void ABY3Engine::share_bools(const vecbool &X_plain, array<vecbool, 2> &Xj) {
    const size_t n = X_plain.size();
    const Peers pe(party_id);
    Dev<int8_t> out(gpu, n);
    {
        Dev<int8_t> x(gpu, X_plain.data(), n);
        GPU_OK(gpu, mpcx_aby3_share_bools_local(gpu, party_id == 0 ? x.p : nullptr, n, prgs[pe.next].init_seed,
                                                prgs[pe.next].pops, prgs[pe.prev].init_seed, prgs[pe.prev].pops, out));
    }
    prgs[pe.next].advance(gpu, n);
    prgs[pe.prev].advance(gpu, n);
    message_send[pe.next].clear();
    out.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecbool(message_send[pe.prev], Xj[0]);
    string2vecbool(message_recv[pe.next], Xj[1]);
}

// ---- opening: send share 0 to next, receive the missing share from prev, combine the three
void ABY3Engine::reveal(const array<VecI64, 2> &Xj, VecI64 &X_plain) {
    const size_t n = Xj[0].size();
    const Peers pe(party_id);
    message_send[pe.prev].clear();
    vecI642string(Xj[0], message_send[pe.next]);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.prev].size() == n * sizeof(int64_t));
    Dev<int64_t> a(gpu, n), b(gpu, Xj[0].data(), n), c(gpu, Xj[1].data(), n), out(gpu, n);
    upload_string(a, message_recv[pe.prev]);
    GPU_OK(gpu, mpcx_aby3_reveal3(gpu, a, b, c, out, n));
    X_plain.resize(n);
    out.down(X_plain.data(), n);
}

void ABY3Engine::reveal_matrix(const array<MatrixXl, 2> &Xj, MatrixXl &X_plain) {
    const size_t rows = Xj[0].rows(), cols = Xj[0].cols(), n = rows * cols;
    assert(Xj[1].rows() == rows && Xj[1].cols() == cols);
    const Peers pe(party_id);
    message_send[pe.prev].clear();
    matrix2string(Xj[0], message_send[pe.next]);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.prev].size() == n * sizeof(int64_t));
    string other;
    matrix2string(Xj[1], other);
    Dev<int64_t> a(gpu, n), b(gpu, n), c(gpu, n), out(gpu, n);
    upload_string(a, message_recv[pe.prev]);
    upload_string(b, message_send[pe.next]);
    upload_string(c, other);
    GPU_OK(gpu, mpcx_aby3_reveal3(gpu, a, b, c, out, n));
    out.to_string(other);
    X_plain.resize(rows, cols);
    string2matrix(other, X_plain);
}

void ABY3Engine::reveal_bools(const array<vecbool, 2> &Xj, vecbool &X_plain) {
    const size_t n = Xj[0].size();
    const Peers pe(party_id);
    message_send[pe.prev].clear();
    vecbool2string(Xj[0], message_send[pe.next]);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.prev].size() == n);
    Dev<int8_t> a(gpu, n), b(gpu, Xj[0].data(), n), c(gpu, Xj[1].data(), n), out(gpu, n);
    upload_string(a, message_recv[pe.prev]);
    GPU_OK(gpu, mpcx_aby3_reveal3_bools(gpu, a, b, c, out, n));
    X_plain.resize(n);
    out.down(X_plain.data(), n);
}

void ABY3Engine::reveal_bits(const array<VecI64, 2> &Xj, VecI64 &X_plain) {
    const size_t n = Xj[0].size();
    const Peers pe(party_id);
    message_send[pe.prev].clear();
    vecI642string(Xj[0], message_send[pe.next]);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.prev].size() == n * sizeof(int64_t));
    Dev<int64_t> a(gpu, n), b(gpu, Xj[0].data(), n), c(gpu, Xj[1].data(), n), out(gpu, n);
    upload_string(a, message_recv[pe.prev]);
    GPU_OK(gpu, mpcx_aby3_reveal3_bits(gpu, a, b, c, out, n));
    X_plain.resize(n);
    out.down(X_plain.data(), n);
}

// A row-major message of rows x cols words into a (column-major) matrix.  Large ones go through the device -- upload,
// turn, download: three streaming copies -- because a transposing pass over host memory is several times slower than a
// streaming one; small ones are turned on the host.
static const size_t kDeviceTransposeWords = (size_t)1 << 18;
void ABY3Engine::wire_to_matrix(const string &wire, size_t rows, size_t cols, MatrixXl &M) {
    M.resize(rows, cols);
    if (rows * cols < kDeviceTransposeWords) {
        string2matrix(wire, M);
        return;
    }
    Dev<int64_t> rm(gpu, rows * cols), cm(gpu, rows * cols);
    upload_string(rm, wire);
    GPU_OK(gpu, mpcx_relayout(gpu, rm, cm, rows, cols, 0, 1));
    cm.down(M.data(), rows * cols);
}

// ---- products
void ABY3Engine::matrix_mul(const array<MatrixXl, 2> &Xj, const array<MatrixXl, 2> &Yj, array<MatrixXl, 2> &Zj) {
    const Peers pe(party_id);
    const size_t rows = Xj[0].rows(), lens = Xj[0].cols(), cols = Yj[0].cols();
    assert(Yj[0].rows() == lens);
    assert(Xj[1].rows() == rows && Xj[1].cols() == lens);
    assert(Yj[1].rows() == lens && Yj[1].cols() == cols);
    message_send[pe.next].clear();
    bool z0_done = false; // Zj[0] already written from the device copy of the product
    if (group && rows && lens && cols) {
        // row blocks over the group's GPUs, each written into its place of the outgoing message; a pinned second operand
        // is on every member already
        message_send[pe.prev].resize(rows * cols * sizeof(int64_t));
        const Pinned *py = find_pinned(Yj);
        assert(!py || (py->rows == lens && py->cols == cols));
        const int rc = py ? mpcx_group_aby3_matmul_local_resident(group, Xj[0].data(), Xj[1].data(), py->gop, rows, lens, cols, 1,
                                                                  prgs[pe.next].init_seed, prgs[pe.next].pops,
                                                                  prgs[pe.prev].init_seed, prgs[pe.prev].pops,
                                                                  (int64_t *)&message_send[pe.prev][0])
                          : mpcx_group_aby3_matmul_local(group, Xj[0].data(), Xj[1].data(), Yj[0].data(), Yj[1].data(), rows,
                                                         lens, cols, 1, prgs[pe.next].init_seed, prgs[pe.next].pops,
                                                         prgs[pe.prev].init_seed, prgs[pe.prev].pops,
                                                         (int64_t *)&message_send[pe.prev][0]);
        if (rc != 0) {
            fprintf(stderr, "mpcx_group_aby3_matmul_local: %s\n", mpcx_group_last_error(group));
            abort();
        }
    } else {
        Dev<int64_t> z(gpu, rows * cols);
        std::unique_ptr<mpcgpu::Span> up(new mpcgpu::Span("matrix_mul: upload"));
        // operands the caller pinned are on the device already
        const Pinned *px = find_pinned(Xj);
        Pinned *py = find_pinned(Yj);
        assert(!px || (px->rows == rows && px->cols == lens));
        assert(!py || (py->rows == lens && py->cols == cols));
        std::unique_ptr<Dev<int64_t>> ux0, ux1, uy0, uy1;
        if (!px) {
            ux0.reset(new Dev<int64_t>(gpu, Xj[0].data(), rows * lens));
            ux1.reset(new Dev<int64_t>(gpu, Xj[1].data(), rows * lens));
        }
        if (!py) {
            uy0.reset(new Dev<int64_t>(gpu, Yj[0].data(), lens * cols));
            uy1.reset(new Dev<int64_t>(gpu, Yj[1].data(), lens * cols));
        }
        const int64_t *x0 = px ? px->s0->p : ux0->p, *x1 = px ? px->s1->p : ux1->p;
        const int64_t *y0 = py ? py->s0->p : uy0->p, *y1 = py ? py->s1->p : uy1->p;
        if (py && !py->as_second) GPU_OK(gpu, mpcx_aby3_operand_prepare(gpu, y0, y1, lens, cols, 1, &py->as_second));
        up.reset();
        // Eigen stores column-major (layout 1); the result comes back row-major = the wire order
        {
            mpcgpu::Span s2("matrix_mul: product");
            if (py)
                GPU_OK(gpu, mpcx_aby3_matmul_local_prepared(gpu, x0, x1, 1, py->as_second, rows, 0, prgs[pe.next].init_seed,
                                                            prgs[pe.next].pops, prgs[pe.prev].init_seed, prgs[pe.prev].pops, z));
            else
                GPU_OK(gpu, mpcx_aby3_matmul_local(gpu, x0, x1, y0, y1, rows, lens, cols, 1, prgs[pe.next].init_seed,
                                                   prgs[pe.next].pops, prgs[pe.prev].init_seed, prgs[pe.prev].pops, z));
            if (mpcgpu::Span::on()) GPU_OK(gpu, mpcx_sync(gpu)); // so that the span is the product's own time
        }
        mpcgpu::Span s3("matrix_mul: download");
        z.to_string(message_send[pe.prev]);
        if (rows * cols >= kDeviceTransposeWords) {
            // the caller's copy of the share, column-major as Eigen stores it: turned on the device and copied straight
            // into the matrix (a transposing pass over host memory costs several such copies)
            Dev<int64_t> zc(gpu, rows * cols);
            GPU_OK(gpu, mpcx_relayout(gpu, z, zc, rows, cols, 0, 1));
            Zj[0].resize(rows, cols);
            zc.down(Zj[0].data(), rows * cols);
            z0_done = true;
        }
    }
    {
        mpcgpu::Span s4("matrix_mul: advance");
        prgs[pe.next].advance(gpu, rows * cols);
        prgs[pe.prev].advance(gpu, rows * cols);
    }
    {
        mpcgpu::Span s5("matrix_mul: round");
        messenger->send_and_recv(message_send, message_recv);
    }
    mpcgpu::Span s6("matrix_mul: to matrices");
    MSG_CHECK(message_recv[pe.next].size() == rows * cols * sizeof(int64_t));
    if (!z0_done) wire_to_matrix(message_send[pe.prev], rows, cols, Zj[0]);
    wire_to_matrix(message_recv[pe.next], rows, cols, Zj[1]);
#if defined MYDEBUG // the product of the revealed operands (ABY3Engine.cpp:202-209)
    MatrixXl xd, yd, zd;
    reveal_matrix(Xj, xd);
    reveal_matrix(Yj, yd);
    reveal_matrix(Zj, zd);
    assert(zd == xd * yd);
#endif
}

void ABY3Engine::vector_mul(const array<VecI64, 2> &Xj, const array<VecI64, 2> &Yj, array<VecI64, 2> &Zj) {
    const Peers pe(party_id);
    const size_t n = Xj[0].size();
    assert((size_t)Xj[1].size() == n && (size_t)Yj[0].size() == n && (size_t)Yj[1].size() == n);
    Dev<int64_t> z(gpu, n);
    {
        Dev<int64_t> x0(gpu, Xj[0].data(), n), x1(gpu, Xj[1].data(), n), y0(gpu, Yj[0].data(), n), y1(gpu, Yj[1].data(), n);
        GPU_OK(gpu, mpcx_aby3_vecmul_local(gpu, x0, x1, y0, y1, n, prgs[pe.next].init_seed, prgs[pe.next].pops,
                                           prgs[pe.prev].init_seed, prgs[pe.prev].pops, z));
    }
    prgs[pe.next].advance(gpu, n);
    prgs[pe.prev].advance(gpu, n);
    message_send[pe.next].clear();
    z.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecI64(message_send[pe.prev], Zj[0]);
    string2vecI64(message_recv[pe.next], Zj[1]);
#if defined MYDEBUG // :231-241
    VecI64 xd, yd, zd;
    reveal(Xj, xd);
    reveal(Yj, yd);
    reveal(Zj, zd);
    for (size_t i = 0; i < (size_t)zd.size(); i++) assert((uint64_t)zd[i] == (uint64_t)xd[i] * (uint64_t)yd[i]);
#endif
}

void ABY3Engine::vecbool_mul(const array<vecbool, 2> &Xj, const array<vecbool, 2> &Yj, array<vecbool, 2> &Zj) {
    const Peers pe(party_id);
    const size_t n = Xj[0].size();
    assert(Xj[1].size() == n && Yj[0].size() == n && Yj[1].size() == n);
    Dev<int8_t> z(gpu, n);
    {
        Dev<int8_t> x0(gpu, Xj[0].data(), n), x1(gpu, Xj[1].data(), n), y0(gpu, Yj[0].data(), n), y1(gpu, Yj[1].data(), n);
        GPU_OK(gpu, mpcx_aby3_vecbool_mul_local(gpu, x0, x1, y0, y1, n, z));
    }
    message_send[pe.next].clear();
    z.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecbool(message_send[pe.prev], Zj[0]);
    string2vecbool(message_recv[pe.next], Zj[1]);
#if defined MYDEBUG // :261-270
    vecbool xd, yd, zd;
    reveal_bools(Xj, xd);
    reveal_bools(Yj, yd);
    reveal_bools(Zj, zd);
    for (size_t i = 0; i < zd.size(); i++) assert(zd[i] == (xd[i] & yd[i]));
#endif
}

// ---- carry-save step before msb: two rounds
void ABY3Engine::add_prepare(const array<VecI64, 2> &Xj, array<VecI64, 2> &Gj, array<VecI64, 2> &Pj) {
    const Peers pe(party_id);
    const size_t n = Xj[0].size();
    assert((size_t)Xj[1].size() == n);
    if (group) return add_prepare_group(*this, Xj, Gj, Pj);
    Dev<int64_t> x0(gpu, Xj[0].data(), n), x1(gpu, Xj[1].data(), n), c0(gpu, n), c1(gpu, n);
    GPU_OK(gpu, mpcx_aby3_add_prepare_step1(gpu, x0, x1, n, c0));
    message_send[pe.next].clear();
    c0.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.next].size() == n * sizeof(int64_t));
    upload_string(c1, message_recv[pe.next]);
    Dev<int64_t> p0(gpu, n), p1(gpu, n), g0(gpu, n);
    GPU_OK(gpu, mpcx_aby3_add_prepare_step2(gpu, x0, x1, c0, c1, n, p0, p1, g0));
    message_send[pe.next].clear();
    g0.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecI64(message_send[pe.prev], Gj[0]);
    string2vecI64(message_recv[pe.next], Gj[1]);
    Pj[0].resize(n);
    Pj[1].resize(n);
    p0.down(Pj[0].data(), n);
    p1.down(Pj[1].data(), n);
#if defined MYDEBUG
    // :327-345 asserts S + 2C = X, P = S ^ 2C, G = S & 2C on the revealed words; with S and 2C internal to the
    // device segments the same statement reads X = (S ^ 2C) + 2 (S & 2C) = P + 2 G
    VecI64 xd, gd, pd;
    reveal(Xj, xd);
    reveal_bits(Gj, gd);
    reveal_bits(Pj, pd);
    for (size_t i = 0; i < (size_t)xd.size(); i++) assert((uint64_t)xd[i] == (uint64_t)pd[i] + 2 * (uint64_t)gd[i]);
#endif
}

// ---- most significant bit of the sum: 6 rounds of the carry tree
void ABY3Engine::msb(const array<VecI64, 2> &Gj, const array<VecI64, 2> &Pj, array<vecbool, 2> &Zj) {
    const Peers pe(party_id);
    const size_t n = Gj[0].size();
    assert(n == (size_t)Gj[1].size() && n == (size_t)Pj[0].size() && n == (size_t)Pj[1].size());
    if (group) return msb_group(*this, Gj, Pj, Zj);
    Dev<int64_t> p0(gpu, Pj[0].data(), n), p1(gpu, Pj[1].data(), n);
    Dev<uint64_t> state(gpu, 6 * n);
    Dev<int64_t> g0(gpu, Gj[0].data(), n), g1(gpu, Gj[1].data(), n); // read in place by the first round
    GPU_OK(gpu, mpcx_aby3_msb_begin(gpu, g0, g1, p0, p1, n, state));
    // one device call per local segment: the first message, then for every received message the next one, then Z
    Dev<int8_t> msg_in(gpu, 64 * n), msg_out(gpu, 64 * n), z0(gpu, n), z1(gpu, n);
    msb_first(gpu, state, n, msg_out);
    for (size_t temp = 32; temp != 0; temp /= 2) {
        const size_t bytes = 2 * temp * n;
        message_send[pe.next].clear();
        message_send[pe.prev].resize(bytes);
        if (bytes) msb_msg_down(gpu, &message_send[pe.prev][0], msg_out, bytes);
        messenger->send_and_recv(message_send, message_recv);
        MSG_CHECK(message_recv[pe.next].size() == bytes);
        msb_msg_up(gpu, msg_in, message_recv[pe.next].data(), bytes);
        if (temp != 1)
            msb_next(gpu, state, n, temp, msg_in, msg_out);
        else
            msb_end(gpu, state, n, msg_in, z0, z1);
    }
    Zj[0].resize(n);
    Zj[1].resize(n);
    z0.down(Zj[0].data(), n);
    z1.down(Zj[1].data(), n);
#if defined MYDEBUG
    // the carry tree's result: the sign bit of P + 2 G, the value add_prepare split (the per-level statements of
    // :430-444 hold inside the device segments, which are checked level by level against the CPU restatement)
    VecI64 gd, pd;
    vecbool zd;
    reveal_bits(Gj, gd);
    reveal_bits(Pj, pd);
    reveal_bools(Zj, zd);
    for (size_t i = 0; i < zd.size(); i++) assert(zd[i] == (int8_t)(((uint64_t)pd[i] + 2 * (uint64_t)gd[i]) >> 63));
#endif
}

void ABY3Engine::trunc(const array<VecI64, 2> &Xj, array<VecI64, 2> &Yj, size_t number) {
    assert(number < 64);
    const Peers pe(party_id);
    const size_t n = Xj[0].size();
    assert(n == (size_t)Xj[1].size());
    // parties 1 and 2 draw from the stream they share: prgs[2] on party 1, prgs[1] on party 2
    const size_t k = party_id == 1 ? 2 : 1;
    // party 0 shifts its first share, party 1 the sum of its two, party 2 only draws from the stream (:480-503): each
    // party uploads what its segment reads
    Dev<int64_t> y(gpu, n);
    std::unique_ptr<Dev<int64_t>> x0, x1;
    if (party_id != 2) x0.reset(new Dev<int64_t>(gpu, Xj[0].data(), n));
    if (party_id == 1) x1.reset(new Dev<int64_t>(gpu, Xj[1].data(), n));
    GPU_OK(gpu, mpcx_aby3_trunc_local(gpu, (int)party_id, x0 ? x0->p : nullptr, x1 ? x1->p : nullptr, n, (int)number,
                                      prgs[k].init_seed, prgs[k].pops, y));
    if (party_id != 0) prgs[k].advance(gpu, n);
    message_send[pe.next].clear();
    y.to_string(message_send[pe.prev]);
    messenger->send_and_recv(message_send, message_recv);
    string2vecI64(message_send[pe.prev], Yj[0]);
    string2vecI64(message_recv[pe.next], Yj[1]);
}

// ---- C = A * b for an arithmetic sharing A and a boolean sharing b: two three-party OTs in one round
void ABY3Engine::bit_inject(const array<VecI64, 2> &Aj, const array<vecbool, 2> &Bj, array<VecI64, 2> &Cj) {
    const size_t n = Aj[0].size();
    assert(n == (size_t)Aj[1].size() && n == Bj[0].size() && n == Bj[1].size());
    if (group) return bit_inject_group(*this, Aj, Bj, Cj);
    const size_t w = n * sizeof(int64_t);
    for (auto &m : message_send) m.clear();
    // the OT senders read a (party 0: its first share, party 1: both), party 2 only the bit shares: each party uploads
    // what its segments read
    Dev<int64_t> a0(gpu, party_id != 2 ? n : 0), a1(gpu, party_id == 1 ? n : 0), c0(gpu, n), c1(gpu, n);
    if (party_id != 2) a0.up(Aj[0].data(), n);
    if (party_id == 1) a1.up(Aj[1].data(), n);
    Dev<int8_t> b0(gpu, Bj[0].data(), n), b1(gpu, Bj[1].data(), n);
    if (party_id == 0) {
        PRgenerator &s01 = prgs[1], &s20 = prgs[2];
        const uint64_t pop01 = s01.pops, pop20 = s20.pops;
        Dev<int64_t> to1(gpu, 2 * n), to2(gpu, 3 * n), from1(gpu, 2 * n), from2(gpu, n);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p0_pre(gpu, a0, b0, b1, n, 0, n, s01.init_seed, pop01, s20.init_seed, pop20, to1, to2));
        s01.advance(gpu, 6 * n);
        s20.advance(gpu, 3 * n);
        to1.to_string(message_send[1]);
        to2.to_string(message_send[2]);
        messenger->send_and_recv(message_send, message_recv);
        MSG_CHECK(message_recv[1].size() == 2 * w && message_recv[2].size() == w);
        upload_string(from1, message_recv[1]);
        upload_string(from2, message_recv[2]);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p0_post(gpu, b0, n, 0, n, s01.init_seed, pop01, s20.init_seed, pop20, from1, from2, c0, c1));
    } else if (party_id == 1) {
        PRgenerator &s01 = prgs[0], &s12 = prgs[2];
        const uint64_t pop01 = s01.pops, pop12 = s12.pops;
        Dev<int64_t> to0(gpu, 2 * n), to2(gpu, 3 * n), from0(gpu, 2 * n), from2(gpu, n);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p1_pre(gpu, a0, a1, b0, b1, n, 0, n, s01.init_seed, pop01, s12.init_seed, pop12, to0, to2));
        s01.advance(gpu, 6 * n);
        s12.advance(gpu, 3 * n);
        to0.to_string(message_send[0]);
        to2.to_string(message_send[2]);
        messenger->send_and_recv(message_send, message_recv);
        MSG_CHECK(message_recv[0].size() == 2 * w && message_recv[2].size() == w);
        upload_string(from0, message_recv[0]);
        upload_string(from2, message_recv[2]);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p1_post(gpu, b1, n, 0, n, s01.init_seed, pop01, s12.init_seed, pop12, from0, from2, c0, c1));
    } else {
        PRgenerator &s20 = prgs[0], &s12 = prgs[1];
        const uint64_t pop20 = s20.pops, pop12 = s12.pops;
        Dev<int64_t> to0(gpu, n), to1(gpu, n), from0(gpu, 3 * n), from1(gpu, 3 * n);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p2_pre(gpu, b0, b1, n, 0, n, s20.init_seed, pop20, s12.init_seed, pop12, to0, to1));
        s20.advance(gpu, 3 * n);
        s12.advance(gpu, 3 * n);
        to0.to_string(message_send[0]);
        to1.to_string(message_send[1]);
        messenger->send_and_recv(message_send, message_recv);
        MSG_CHECK(message_recv[0].size() == 3 * w && message_recv[1].size() == 3 * w);
        upload_string(from0, message_recv[0]);
        upload_string(from1, message_recv[1]);
        GPU_OK(gpu, mpcx_aby3_bit_inject_p2_post(gpu, b0, b1, n, 0, n, s20.init_seed, pop20, s12.init_seed, pop12, from0, from1, c0, c1));
    }
    Cj[0].resize(n);
    Cj[1].resize(n);
    c0.down(Cj[0].data(), n);
    c1.down(Cj[1].data(), n);
#if defined MYDEBUG // the reference checks the two OT outputs one by one (:1018-1037); their sum is C = A * b
    VecI64 ad, cd;
    vecbool bd;
    reveal(Aj, ad);
    reveal_bools(Bj, bd);
    reveal(Cj, cd);
    for (size_t i = 0; i < (size_t)cd.size(); i++) assert(cd[i] == (bd[i] ? ad[i] : 0));
#endif
}

// ---- ReLU as one protocol.  Written over element ranges (mpcgpu::run): the whole vector on the engine's one GPU, or
// every member of the group its own range.  The rounds, the bytes of every message and the generator words are those
// of add_prepare (:295-326), msb (:361-467) and bit_inject (:510-1038) called one after the other; the last segment is
// bit_inject's fused with the subtraction R = X - C.
void ABY3Engine::relu(const VecPair &Xj, VecPair &Rj) {
    using mpcgpu::Bufs;
    using mpcgpu::fresh;
    const Peers pe(party_id);
    const size_t n = Xj[0].size(), G = mpcgpu::members(*this), id = party_id;
    assert((size_t)Xj[1].size() == n);
    Bufs<int64_t> x0(G), x1(G), c0(G), c1(G), p0(G), p1(G), g0(G), g1(G);
    Bufs<uint64_t> state(G);
    Bufs<int8_t> msg_in(G), msg_out(G), z0(G), z1(G);
    string &out = message_send[pe.prev];
    // add_prepare, round 1: C0 = X0 & X1
    message_send[pe.next].clear();
    out.resize(n * W8);
    mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        fresh(x0[m], ctx, cnt).up(Xj[0].data() + i0, cnt);
        fresh(x1[m], ctx, cnt).up(Xj[1].data() + i0, cnt);
        GPU_OK(ctx, mpcx_aby3_add_prepare_step1(ctx, *x0[m], *x1[m], cnt, fresh(c0[m], ctx, cnt)));
        mpcgpu::put(*c0[m], out, 0, i0, cnt);
    });
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.next].size() == n * W8);
    // round 2: G0
    mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        mpcgpu::get(fresh(c1[m], ctx, cnt), message_recv[pe.next], 0, i0, cnt);
        fresh(p0[m], ctx, cnt);
        fresh(p1[m], ctx, cnt);
        fresh(g0[m], ctx, cnt);
        GPU_OK(ctx, mpcx_aby3_add_prepare_step2(ctx, *x0[m], *x1[m], *c0[m], *c1[m], cnt, *p0[m], *p1[m], *g0[m]));
        mpcgpu::put(*g0[m], out, 0, i0, cnt);
        c0[m].reset();
        c1[m].reset();
    });
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[pe.next].size() == n * W8);
    // msb: the first message from G, P where they lie; then one call per received message
    out.resize(64 * n);
    mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        mpcgpu::get(fresh(g1[m], ctx, cnt), message_recv[pe.next], 0, i0, cnt);
        fresh(state[m], ctx, 6 * cnt);
        fresh(msg_in[m], ctx, 64 * cnt);
        fresh(msg_out[m], ctx, 64 * cnt);
        fresh(z0[m], ctx, cnt);
        fresh(z1[m], ctx, cnt);
        GPU_OK(ctx, mpcx_aby3_msb_begin(ctx, *g0[m], *g1[m], *p0[m], *p1[m], cnt, *state[m]));
        msb_first(ctx, *state[m], cnt, *msg_out[m]);
        msb_msg_down(ctx, &out[64 * i0], *msg_out[m], 64 * cnt);
    });
    for (size_t temp = 32; temp != 0; temp /= 2) {
        const size_t per = 2 * temp; // message bytes per element of this round
        out.resize(per * n);
        messenger->send_and_recv(message_send, message_recv);
        const string &in = message_recv[pe.next];
        MSG_CHECK(in.size() == per * n);
        if (temp != 1) out.resize(per / 2 * n); // the next message is written in place, slice by slice
        mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
            if (!cnt) return;
            msb_msg_up(ctx, *msg_in[m], in.data() + per * i0, per * cnt);
            if (temp != 1) {
                msb_next(ctx, *state[m], cnt, temp, *msg_in[m], *msg_out[m]);
                msb_msg_down(ctx, &out[per / 2 * i0], *msg_out[m], per / 2 * cnt);
            } else {
                msb_end(ctx, *state[m], cnt, *msg_in[m], *z0[m], *z1[m]);
            }
        });
    }
    // bit_inject on (X, sign bits), its last segment fused with the subtraction
    const size_t a = (id + 1) % 3, b = (id + 2) % 3;
    static const size_t send_sections[3][3] = {{0, 2, 3}, {2, 0, 3}, {1, 1, 0}}; // [from][to], in vectors of n words
    PRgenerator &sA = prgs[id == 0 ? 1 : 0], &sB = prgs[id == 2 ? 1 : 2]; // P0: s01, s20; P1: s01, s12; P2: s20, s12
    const uint64_t popA = sA.pops, popB = sB.pops;
    Bufs<int64_t> toA(G), toB(G), fromA(G), fromB(G), r0(G), r1(G);
    for (auto &m : message_send) m.clear();
    message_send[a].resize(send_sections[id][a] * n * W8);
    message_send[b].resize(send_sections[id][b] * n * W8);
    mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        p0[m].reset(), p1[m].reset(), g0[m].reset(), g1[m].reset(), state[m].reset(), msg_in[m].reset(), msg_out[m].reset();
        fresh(toA[m], ctx, send_sections[id][a] * cnt);
        fresh(toB[m], ctx, send_sections[id][b] * cnt);
        if (id == 0)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p0_pre(ctx, *x0[m], *z0[m], *z1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toA[m], *toB[m]));
        else if (id == 1)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p1_pre(ctx, *x0[m], *x1[m], *z0[m], *z1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toB[m], *toA[m]));
        else
            GPU_OK(ctx, mpcx_aby3_bit_inject_p2_pre(ctx, *z0[m], *z1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *toA[m], *toB[m]));
        sections_down(*toA[m], message_send[a], send_sections[id][a], n, i0, cnt);
        sections_down(*toB[m], message_send[b], send_sections[id][b], n, i0, cnt);
    });
    sA.advance(gpu, (id == 2 ? 3 : 6) * n);
    sB.advance(gpu, 3 * n);
    messenger->send_and_recv(message_send, message_recv);
    MSG_CHECK(message_recv[a].size() == send_sections[a][id] * n * W8 && message_recv[b].size() == send_sections[b][id] * n * W8);
    Rj[0].resize(n);
    Rj[1].resize(n);
    mpcgpu::run(*this, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) {
        fresh(fromA[m], ctx, send_sections[a][id] * cnt);
        fresh(fromB[m], ctx, send_sections[b][id] * cnt);
        fresh(r0[m], ctx, cnt);
        fresh(r1[m], ctx, cnt);
        sections_up(ctx, *fromA[m], message_recv[a], send_sections[a][id], n, i0, cnt);
        sections_up(ctx, *fromB[m], message_recv[b], send_sections[b][id], n, i0, cnt);
        if (id == 0)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p0_post_sub(ctx, *z0[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromA[m], *fromB[m], *x0[m], *x1[m], *r0[m], *r1[m]));
        else if (id == 1)
            GPU_OK(ctx, mpcx_aby3_bit_inject_p1_post_sub(ctx, *z1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromB[m], *fromA[m], *x0[m], *x1[m], *r0[m], *r1[m]));
        else
            GPU_OK(ctx, mpcx_aby3_bit_inject_p2_post_sub(ctx, *z0[m], *z1[m], n, i0, cnt, sA.init_seed, popA, sB.init_seed, popB, *fromA[m], *fromB[m], *x0[m], *x1[m], *r0[m], *r1[m]));
        if (cnt) {
            r0[m]->down(Rj[0].data() + i0, cnt);
            r1[m]->down(Rj[1].data() + i0, cnt);
        }
    });
#if defined MYDEBUG // the opened result is max(x, 0)
    VecI64 xd, rd;
    reveal(Xj, xd);
    reveal(Rj, rd);
    for (size_t i = 0; i < (size_t)rd.size(); i++) assert(rd[i] == (xd[i] < 0 ? 0 : xd[i]));
#endif
}
