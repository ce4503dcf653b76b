// Small helpers the host engines use to drive the GPU library (include/mpcx.h): one context per engine,
// RAII device buffers, and the reference's error convention (no exceptions, no return codes: a failed
// call prints the library's message and aborts, like a failed assert).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/mpcx.h"

#define GPU_OK(ctx, call)                                                              \
    do {                                                                               \
        if ((call) != 0) {                                                             \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, mpcx_last_error(ctx)); \
            abort();                                                                   \
        }                                                                              \
    } while (0)

// A received message of the wrong length is a protocol error: checked in every build (an assert would vanish with
// -DNDEBUG and the upload would then run past the device buffer or the host string).
#define MSG_CHECK(cond)                                                                          \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            fprintf(stderr, "%s:%d: unexpected message length: %s\n", __FILE__, __LINE__, #cond); \
            abort();                                                                             \
        }                                                                                        \
    } while (0)

namespace mpcgpu {

#if defined MYDEBUG
// Engines built with -DMYDEBUG (the reference's switch, mpc_engine/ABY3Engine.cpp:3) open their operands and results
// after a protocol and assert the relation it computes.  The checks cost extra messenger rounds, so every party has
// to be built the same way.  A method with several exits declares its check first; it runs when the method returns.
template <class F> struct AtExit {
    F f;
    explicit AtExit(F fn) : f(fn) {}
    AtExit(const AtExit &) = delete;
    ~AtExit() { f(); }
};
#endif

// MPCX_ENGINE_TRACE=1: wall time of the steps of an engine method on stderr (one line per step and call)
struct Span {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    static bool on() {
        static const bool v = [] {
            const char *e = getenv("MPCX_ENGINE_TRACE");
            return e && atoi(e) != 0;
        }();
        return v;
    }
    explicit Span(const char *w) : what(w) {
        if (on()) t0 = std::chrono::steady_clock::now();
    }
    ~Span() {
        if (on())
            fprintf(stderr, "[engine] %-28s %9.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

inline mpcx_ctx *open_context() {
    mpcx_ctx *ctx = nullptr;
    const char *dev = getenv("MPCX_DEVICE");
    if (mpcx_init(dev ? atoi(dev) : 0, &ctx) != 0) {
        fprintf(stderr, "mpcx_init: %s\n", mpcx_last_error(nullptr));
        abort();
    }
    return ctx;
}

// MPCX_DEVICES=a,b,..: the GPUs one engine spreads its work over (NULL when fewer than two are listed)
inline mpcx_group *open_group() {
    const char *list = getenv("MPCX_DEVICES");
    if (!list) return nullptr;
    std::vector<int> devs;
    for (const char *p = list; *p;) {
        char *end;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        devs.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
    }
    if (devs.size() < 2) return nullptr;
    mpcx_group *g = nullptr;
    if (mpcx_group_init(devs.data(), (int)devs.size(), &g) != 0) {
        fprintf(stderr, "mpcx_group_init: %s\n", mpcx_last_error(nullptr));
        abort();
    }
    return g;
}
// f(ctx, member, first, count) for every member's range of [0, n), on the members' threads
template <class F> inline void group_for_each(mpcx_group *g, uint64_t n, uint64_t granule, F f) {
    struct Thunk {
        static int call(mpcx_ctx *ctx, int member, uint64_t first, uint64_t count, void *user) {
            (*(F *)user)(ctx, member, first, count);
            return 0;
        }
    };
    if (mpcx_group_for_each(g, n, granule, &Thunk::call, &f) != 0) {
        fprintf(stderr, "mpcx_group_for_each: %s\n", mpcx_group_last_error(g));
        abort();
    }
}

// typed device buffer owned by this object
template <class T> class Dev {
  public:
    mpcx_ctx *ctx;
    T *p = nullptr;
    size_t n = 0;
    Dev(mpcx_ctx *c, size_t count) : ctx(c), n(count) {
        void *q = nullptr;
        GPU_OK(ctx, mpcx_malloc(ctx, (count ? count : 1) * sizeof(T), &q));
        p = (T *)q;
    }
    Dev(mpcx_ctx *c, const T *host, size_t count) : Dev(c, count) { up(host, count); }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    ~Dev() { mpcx_free(ctx, p); }
    // 16-bit arrays are the Z_67 shares of privateCompare (values below 67): the library moves such an array over the
    // link as bytes when its values allow it, and as it is otherwise
    void up(const void *host, size_t count) {
        if (sizeof(T) == 2)
            GPU_OK(ctx, mpcx_h2d_u16(ctx, p, host, count));
        else
            GPU_OK(ctx, mpcx_h2d(ctx, p, host, count * sizeof(T)));
    }
    void down(void *host, size_t count, size_t first = 0) const {
        if (sizeof(T) == 2)
            GPU_OK(ctx, mpcx_d2h_u16(ctx, host, p + first, count));
        else
            GPU_OK(ctx, mpcx_d2h(ctx, host, p + first, count * sizeof(T)));
    }
    // the whole buffer as a wire payload
    void to_string(std::string &s) const {
        s.resize(n * sizeof(T));
        if (n) down(&s[0], n);
    }
    operator T *() { return p; }
    operator const T *() const { return p; }
};

// bytes [byte_off, byte_off + bytes) of a message into the buffer from item dev_first on
template <class T> inline void upload_string_at(Dev<T> &d, size_t dev_first, const std::string &s, size_t byte_off, size_t bytes) {
    if (byte_off > s.size() || bytes > s.size() - byte_off) {
        fprintf(stderr, "upload_string_at: bytes [%zu, +%zu) outside a message of %zu bytes\n", byte_off, bytes, s.size());
        abort();
    }
    if (dev_first > d.n || bytes > (d.n - dev_first) * sizeof(T)) {
        fprintf(stderr, "upload_string_at: %zu bytes from item %zu do not fit a device buffer of %zu items\n", bytes, dev_first,
                d.n);
        abort();
    }
    if (!bytes) return;
    if (sizeof(T) == 2 && bytes % 2 == 0)
        GPU_OK(d.ctx, mpcx_h2d_u16(d.ctx, d.p + dev_first, s.data() + byte_off, bytes / 2));
    else
        GPU_OK(d.ctx, mpcx_h2d(d.ctx, d.p + dev_first, s.data() + byte_off, bytes));
}

template <class T> inline void upload_string(Dev<T> &d, const std::string &s, size_t byte_off = 0, size_t bytes = (size_t)-1) {
    if (byte_off > s.size() || (bytes != (size_t)-1 && bytes > s.size() - byte_off)) {
        fprintf(stderr, "upload_string: bytes [%zu, +%zu) outside a message of %zu bytes\n", byte_off, bytes, s.size());
        abort();
    }
    if (bytes == (size_t)-1) bytes = s.size() - byte_off;
    if (bytes > d.n * sizeof(T)) {
        fprintf(stderr, "upload_string: %zu bytes do not fit a device buffer of %zu bytes\n", bytes, d.n * sizeof(T));
        abort();
    }
    if (!bytes) return;
    if (sizeof(T) == 2 && bytes % 2 == 0)
        GPU_OK(d.ctx, mpcx_h2d_u16(d.ctx, d.p, s.data() + byte_off, bytes / 2));
    else
        GPU_OK(d.ctx, mpcx_h2d(d.ctx, d.p, s.data() + byte_off, bytes));
}

// ---- protocols written over element ranges.  An engine E has a context `gpu` and a group `group` (NULL without
// MPCX_DEVICES).  run() hands the whole vector to the engine's one GPU, or every member of the group its own contiguous
// range [first, first + count) on the member's thread.  A member keeps its device buffers across the rounds (one slot
// per member in a Bufs), runs the range forms of the library's segments and moves its own slices of the messages: a
// message is the concatenation of sections (one per array it carries), each section holding the elements in order.
const uint64_t RANGE_GRANULE = 64;
template <class T> using Buf = std::unique_ptr<Dev<T>>;
template <class T> using Bufs = std::vector<Buf<T>>;

// f(ctx, member, first, count)
template <class E, class F> inline void run(E &e, size_t n, F f) {
    if (e.group && n)
        group_for_each(e.group, n, RANGE_GRANULE, f);
    else
        f(e.gpu, 0, (uint64_t)0, (uint64_t)n);
}
template <class E> inline size_t members(const E &e) { return e.group ? (size_t)mpcx_group_size(e.group) : 1; }
template <class T> inline Dev<T> &fresh(Buf<T> &b, mpcx_ctx *ctx, size_t count) {
    b.reset(new Dev<T>(ctx, count));
    return *b;
}
// `count` elements of `per` items each, from item dev_first of the buffer, into / out of the section that starts at
// byte `section` of the message, at element `first`
template <class T>
inline void put(const Dev<T> &d, std::string &msg, size_t section, size_t first, size_t count, size_t per = 1,
                size_t dev_first = 0) {
    if (count) d.down(&msg[section + first * per * sizeof(T)], count * per, dev_first);
}
template <class T>
inline void get(Dev<T> &d, const std::string &msg, size_t section, size_t first, size_t count, size_t per = 1,
                size_t dev_first = 0) {
    upload_string_at(d, dev_first, msg, section + first * per * sizeof(T), count * per * sizeof(T));
}
// `nsec` consecutive sections of n elements each (the message from byte `section` on) against a buffer that holds the
// member's `count` elements of every section back to back: one copy when the member has the whole vector
template <class T>
inline void put_sections(const Dev<T> &d, std::string &msg, size_t section, size_t nsec, size_t n, size_t first, size_t count) {
    if (count == n) return put(d, msg, section, 0, nsec * n);
    for (size_t k = 0; k < nsec; k++) put(d, msg, section + k * n * sizeof(T), first, count, 1, k * count);
}
template <class T>
inline void get_sections(Dev<T> &d, const std::string &msg, size_t section, size_t nsec, size_t n, size_t first, size_t count) {
    if (count == n) return get(d, msg, section, 0, nsec * n);
    for (size_t k = 0; k < nsec; k++) get(d, msg, section + k * n * sizeof(T), first, count, 1, k * count);
}
// host vector -> one device buffer per member, and back
template <class E, class T> inline void scatter(E &e, const T *host, size_t n, size_t per, Bufs<T> &d) {
    run(e, n, [&](mpcx_ctx *ctx, int m, uint64_t i0, uint64_t cnt) { fresh(d[m], ctx, per * cnt).up(host + per * i0, per * cnt); });
}
template <class E, class T> inline void gather(E &e, const Bufs<T> &d, size_t n, T *host) {
    run(e, n, [&](mpcx_ctx *, int m, uint64_t i0, uint64_t cnt) {
        if (cnt) d[m]->down(host + i0, cnt);
    });
}

} // namespace mpcgpu
