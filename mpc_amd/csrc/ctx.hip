// Context, memory and error plumbing of libmpcx.
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "mpcx_internal.h"

// ---- a small pool of host threads for copies between pageable and pinned memory.  One core copies ~12 GB/s; a
// message of hundreds of megabytes has to move at the rate of the PCIe link (~55 GB/s), so large copies are cut into
// pieces that several threads move at once.  The pool lives until the process ends (its objects are never destroyed,
// so no thread can outlive them).
namespace {
struct CopyPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    unsigned threads = 0;
    CopyPool() {
        // default: three quarters of the hardware threads, between 4 and 16.  The callers wait for the pool while it works,
        // so their own threads do not compete with it; measured with three party threads on a 16-thread host
        // (4096 x 4096 matrix_mul + trunc through the engines): 4 / 8 / 12 / 16 pool threads -> 347 / 178 / 137 / 136 ms.
        const char *e = getenv("MPCX_COPY_THREADS");
        long n = e ? atol(e) : (long)(std::thread::hardware_concurrency() * 3 / 4);
        if (!e) n = n < 4 ? 4 : n > 16 ? 16 : n;
        threads = (unsigned)(n < 0 ? 0 : n > 32 ? 32 : n);
        for (unsigned i = 0; i < threads; i++)
            std::thread([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !jobs.empty(); });
                        job = std::move(jobs.front());
                        jobs.pop_front();
                    }
                    job();
                }
            }).detach();
    }
};
CopyPool &copy_pool() {
    static CopyPool *p = new CopyPool(); // never destroyed
    return *p;
}
} // namespace

// MPCX_COPY_STATS=1: bytes and wall time spent in mpcx_h2d / mpcx_d2h / mpcx_host_copy, printed when the process ends
namespace {
struct CopyStats {
    std::atomic<u64> bytes[3], ns[3], calls[3];
    bool on;
    CopyStats() {
        const char *e = getenv("MPCX_COPY_STATS");
        on = e && atoi(e) != 0;
        for (int k = 0; k < 3; k++) bytes[k] = ns[k] = calls[k] = 0;
        if (on) atexit([] {
            static const char *names[3] = {"h2d", "d2h", "host_copy"};
            CopyStats &S = copy_stats();
            for (int k = 0; k < 3; k++)
                fprintf(stderr, "[mpcx] %-9s %8llu calls %10.1f MB %9.2f ms (summed over threads) %7.2f GB/s\n", names[k],
                        (unsigned long long)S.calls[k].load(), S.bytes[k].load() / 1e6, S.ns[k].load() / 1e6,
                        S.ns[k].load() ? (double)S.bytes[k].load() / (double)S.ns[k].load() : 0.0);
        });
    }
    static CopyStats &copy_stats() {
        static CopyStats *s = new CopyStats();
        return *s;
    }
};
struct CopyTimer {
    int slot;
    u64 bytes;
    std::chrono::steady_clock::time_point t0;
    CopyTimer(int s, u64 b) : slot(s), bytes(b) {
        if (CopyStats::copy_stats().on) t0 = std::chrono::steady_clock::now();
    }
    ~CopyTimer() {
        CopyStats &S = CopyStats::copy_stats();
        if (!S.on) return;
        S.ns[slot] += (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        S.bytes[slot] += bytes;
        S.calls[slot]++;
    }
};
} // namespace

// fn(k) for k in [0, pieces): piece 0 on the calling thread, the others on the pool
template <class F> static void pool_run(u64 pieces, F fn) {
    CopyPool &P = copy_pool();
    if (pieces <= 1 || P.threads == 0) {
        for (u64 k = 0; k < pieces; k++) fn(k);
        return;
    }
    struct Latch {
        std::mutex mu;
        std::condition_variable cv;
        u64 left;
    };
    // shared with the jobs: the last one may still be inside its unlock when this function returns
    auto latch = std::make_shared<Latch>();
    latch->left = pieces - 1;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (u64 k = 1; k < pieces; k++)
            P.jobs.emplace_back([=] {
                fn(k);
                std::lock_guard<std::mutex> g(latch->mu);
                if (--latch->left == 0) latch->cv.notify_one();
            });
    }
    P.cv.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(latch->mu);
    latch->cv.wait(lk, [&] { return latch->left == 0; });
}

void mpcx_parallel_memcpy(void *dst, const void *src, u64 bytes) {
    CopyPool &P = copy_pool();
    const u64 piece_min = 1ull << 20;
    if (P.threads == 0 || bytes < 2 * piece_min) {
        memcpy(dst, src, bytes);
        return;
    }
    u64 pieces = bytes / piece_min;
    if (pieces > P.threads + 1) pieces = P.threads + 1; // the caller copies a piece itself
    const u64 per = ((bytes + pieces - 1) / pieces + 63) & ~63ull;
    pool_run(pieces, [=](u64 k) {
        const u64 off = k * per;
        if (off < bytes) memcpy((char *)dst + off, (const char *)src + off, bytes - off < per ? bytes - off : per);
    });
}

// dst (C x R, row-major) = transpose of src (R x C, row-major), 8-byte words, in 32 x 32 tiles so that neither side is
// walked with a large stride; bands of destination rows go to the pool's threads
static void transpose64_band(u64 *dst, const u64 *src, u64 R, u64 C, u64 c_lo, u64 c_hi) {
    const u64 T = 32;
    for (u64 c0 = c_lo; c0 < c_hi; c0 += T)
        for (u64 r0 = 0; r0 < R; r0 += T) {
            const u64 c1 = c0 + T < c_hi ? c0 + T : c_hi, r1 = r0 + T < R ? r0 + T : R;
            for (u64 c = c0; c < c1; c++)
                for (u64 r = r0; r < r1; r++) dst[c * R + r] = src[r * C + c];
        }
}

static thread_local std::string g_init_err;

int mpcx_fail(mpcx_ctx *ctx, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_init_err = buf;
    return 1;
}

int mpcx_ws(mpcx_ctx *ctx, u64 bytes, void **out) {
    if (bytes > ctx->ws_bytes) {
        MPCX_HIP(hipSetDevice(ctx->device));
        if (ctx->ws) {
            // kernels of earlier calls may still read the old workspace
            MPCX_HIP(hipStreamSynchronize(ctx->stream));
            MPCX_HIP(hipFree(ctx->ws));
            ctx->ws = nullptr;
            ctx->ws_bytes = 0;
        }
        u64 want = (bytes + (1ull << 20) - 1) & ~((1ull << 20) - 1);
        MPCX_HIP(hipMalloc(&ctx->ws, want));
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return 0;
}

extern "C" {

int mpcx_init(int device, mpcx_ctx **out) {
    mpcx_ctx *ctx = nullptr;
    if (!out) return mpcx_fail(nullptr, "mpcx_init: out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return mpcx_fail(nullptr, "mpcx_init: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return mpcx_fail(nullptr, "mpcx_init: device %d out of range (%d devices)", device, count);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return mpcx_fail(nullptr, "mpcx_init: hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return mpcx_fail(nullptr, "mpcx_init: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                         prop.gcnArchName);
    ctx = new mpcx_ctx();
    ctx->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return mpcx_fail(nullptr, "mpcx_init: stream creation failed: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
    if (ring_gemm_static_scratch_init(device)) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return 1;
    }
    prg_tables_context_opened(device);
    *out = ctx;
    return 0;
}

int mpcx_shutdown(mpcx_ctx *ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (int k = 0; k < 2; k++)
        if (ctx->stage[k]) {
            (void)hipHostFree(ctx->stage[k]);
            (void)hipEventDestroy(ctx->stage_ev[k]);
        }
    prg_rand_cache_clear(ctx);
    for (auto &b : ctx->pool_free) (void)hipFree(b.second);
    for (auto &b : ctx->pool_live) (void)hipFree(b.first);
    for (auto &v : ctx->prof_ev)
        for (auto &p : v) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    prg_tables_context_closed(ctx->device);
    delete ctx;
    return 0;
}

const char *mpcx_last_error(mpcx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

int mpcx_set_stream(mpcx_ctx *ctx, void *hip_stream) {
    MPCX_REQUIRE(ctx, "mpcx_set_stream: the context is NULL (already shut down?)");
    if (ctx->stream == (hipStream_t)hip_stream) return 0; // nothing changes (and the stream may be under capture)
    // work queued on the old stream may still use the workspace, cached slices and recycled blocks
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) MPCX_HIP(hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return 0;
}
void *mpcx_get_stream(mpcx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int mpcx_sync(mpcx_ctx *ctx) {
    MPCX_REQUIRE(ctx, "mpcx_sync: the context is NULL (already shut down?)");
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
static int pool_drain(mpcx_ctx *ctx) {
    if (ctx->pool_free.empty()) return 0;
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &b : ctx->pool_free) MPCX_HIP(hipFree(b.second));
    ctx->pool_free.clear();
    ctx->pool_idle_bytes = 0;
    return 0;
}
int mpcx_malloc(mpcx_ctx *ctx, uint64_t bytes, void **out) {
    MPCX_REQUIRE(ctx, "mpcx_malloc: the context is NULL (already shut down?)");
    MPCX_REQUIRE(out, "mpcx_malloc: out is NULL");
    const u64 want = ((bytes ? bytes : 8) + 255) & ~255ull;
    // smallest idle block that fits and wastes at most a quarter
    auto it = ctx->pool_free.lower_bound(want);
    if (it != ctx->pool_free.end() && it->first <= want + want / 4) {
        *out = it->second;
        ctx->pool_live[it->second] = it->first;
        ctx->pool_idle_bytes -= it->first;
        ctx->pool_free.erase(it);
        return 0;
    }
    MPCX_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) { // give the idle blocks back to the driver and try once more
        (void)hipGetLastError();
        MPCX_TRY(pool_drain(ctx));
        MPCX_HIP(hipMalloc(out, want));
    }
    ctx->pool_live[*out] = want;
    return 0;
}
int mpcx_free(mpcx_ctx *ctx, void *p) {
    MPCX_REQUIRE(ctx, "mpcx_free: the context is NULL (already shut down?)");
    if (!p) return 0;
    auto it = ctx->pool_live.find(p);
    MPCX_REQUIRE(it != ctx->pool_live.end(), "mpcx_free: %p was not allocated by mpcx_malloc on this context", p);
    const u64 size = it->second;
    ctx->pool_live.erase(it);
    if (ctx->pool_idle_bytes + size > ctx->pool_cap) {
        MPCX_HIP(hipStreamSynchronize(ctx->stream));
        MPCX_HIP(hipFree(p));
        return 0;
    }
    ctx->pool_free.emplace(size, p);
    ctx->pool_idle_bytes += size;
    return 0;
}
int mpcx_set_pool_limit(mpcx_ctx *ctx, uint64_t max_idle_bytes) {
    MPCX_REQUIRE(ctx, "mpcx_set_pool_limit: the context is NULL (already shut down?)");
    ctx->pool_cap = max_idle_bytes;
    if (ctx->pool_idle_bytes > max_idle_bytes) return pool_drain(ctx);
    return 0;
}
// Large transfers between the device and pageable host memory (the std::string payloads of the messenger) go through
// two pinned staging buffers of the context: the DMA engine moves one chunk while host threads copy the previous one
// between its staging buffer and the caller's memory, so the transfer runs at the rate of the slower of the two
// instead of at the single-threaded rate of a pageable hipMemcpy.  MPCX_STAGED_COPY=0 restores the plain path.
static const u64 kStageChunk = 4ull << 20, kStageMin = 4ull << 20;
static bool staged_copies() {
    static const bool on = [] {
        const char *e = getenv("MPCX_STAGED_COPY");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
static int stage_setup(mpcx_ctx *ctx) {
    if (ctx->stage[0]) return 0;
    MPCX_HIP(hipSetDevice(ctx->device));
    for (int k = 0; k < 2; k++) {
        MPCX_HIP(hipHostMalloc(&ctx->stage[k], kStageChunk, hipHostMallocDefault));
        MPCX_HIP(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    return 0;
}
int mpcx_h2d(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    MPCX_REQUIRE(ctx, "mpcx_h2d: the context is NULL (already shut down?)");
    if (!bytes) return 0;
    CopyTimer timer(0, bytes);
    if (bytes >= kStageMin && staged_copies()) {
        MPCX_TRY(stage_setup(ctx));
        u64 k = 0;
        for (u64 off = 0; off < bytes; off += kStageChunk, k++) {
            const u64 len = bytes - off < kStageChunk ? bytes - off : kStageChunk;
            if (k >= 2) MPCX_HIP(hipEventSynchronize(ctx->stage_ev[k & 1])); // its previous chunk has left the buffer
            mpcx_parallel_memcpy(ctx->stage[k & 1], (const char *)src + off, len);
            MPCX_HIP(hipMemcpyAsync((char *)dst + off, ctx->stage[k & 1], len, hipMemcpyHostToDevice, ctx->stream));
            MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
        }
        MPCX_HIP(hipStreamSynchronize(ctx->stream)); // the staging buffers are free again when the call returns
        return 0;
    }
    MPCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MPCX_HIP(hipStreamSynchronize(ctx->stream)); // the host buffer may be pageable and reused at once
    return 0;
}
} // extern "C"
// `height` runs of `width` bytes that lie `spitch` bytes apart in pageable host memory -> one contiguous device buffer
// (a row block of a column-major matrix: `rows` consecutive words in each column).  The runs are gathered into the
// pinned staging buffers by the host threads, whole runs per chunk, and leave from there like mpcx_h2d's chunks; a
// run longer than a staging buffer goes through mpcx_h2d by itself.
int mpcx_h2d_gather(mpcx_ctx *ctx, void *dst, const void *src, u64 spitch, u64 width, u64 height) {
    MPCX_REQUIRE(ctx, "mpcx_h2d_gather: the context is NULL (already shut down?)");
    if (!width || !height) return 0;
    if (width == spitch) return mpcx_h2d(ctx, dst, src, width * height);
    if (width > kStageChunk) {
        for (u64 h = 0; h < height; h++)
            MPCX_TRY(mpcx_h2d(ctx, (char *)dst + h * width, (const char *)src + h * spitch, width));
        return 0;
    }
    CopyTimer timer(0, width * height);
    MPCX_TRY(stage_setup(ctx));
    const u64 per = kStageChunk / width; // runs per chunk
    u64 k = 0;
    for (u64 h0 = 0; h0 < height; h0 += per, k++) {
        const u64 cnt = height - h0 < per ? height - h0 : per;
        if (k >= 2) MPCX_HIP(hipEventSynchronize(ctx->stage_ev[k & 1])); // its previous chunk has left the buffer
        char *stage = (char *)ctx->stage[k & 1];
        const char *from = (const char *)src + h0 * spitch;
        const u64 pieces = cnt * width < (1u << 16) ? 1 : copy_pool().threads + 1;
        const u64 share = (cnt + pieces - 1) / pieces;
        pool_run(pieces, [=](u64 piece) {
            const u64 lo = piece * share, hi = lo + share < cnt ? lo + share : cnt;
            for (u64 h = lo; h < hi; h++) memcpy(stage + h * width, from + h * spitch, width);
        });
        MPCX_HIP(hipMemcpyAsync((char *)dst + h0 * width, stage, cnt * width, hipMemcpyHostToDevice, ctx->stream));
        MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
    }
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" {
int mpcx_d2h(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    MPCX_REQUIRE(ctx, "mpcx_d2h: the context is NULL (already shut down?)");
    if (!bytes) return 0;
    CopyTimer timer(1, bytes);
    if (bytes >= kStageMin && staged_copies()) {
        MPCX_TRY(stage_setup(ctx));
        const u64 chunks = (bytes + kStageChunk - 1) / kStageChunk;
        auto issue = [&](u64 k) -> int {
            const u64 off = k * kStageChunk, len = bytes - off < kStageChunk ? bytes - off : kStageChunk;
            MPCX_HIP(hipMemcpyAsync(ctx->stage[k & 1], (const char *)src + off, len, hipMemcpyDeviceToHost, ctx->stream));
            MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
            return 0;
        };
        MPCX_TRY(issue(0));
        for (u64 k = 0; k < chunks; k++) {
            if (k + 1 < chunks) MPCX_TRY(issue(k + 1)); // into the other buffer, emptied in the previous iteration
            MPCX_HIP(hipEventSynchronize(ctx->stage_ev[k & 1]));
            const u64 off = k * kStageChunk, len = bytes - off < kStageChunk ? bytes - off : kStageChunk;
            mpcx_parallel_memcpy((char *)dst + off, ctx->stage[k & 1], len);
        }
        return 0;
    }
    MPCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
// ---- boolean messages across the link as bit strings.  The wire carries one byte per boolean (utils/util.hpp:
// vecbool2string); the device keeps booleans as bits.  These two move the bits and make / read the bytes on the host, in
// the copy threads that would otherwise only copy them: an eighth of the bytes cross the link.
static void expand_bits(u8 *dst, const u8 *bits, u64 first_bit, u64 nbits) { // first_bit a multiple of 8
    const u8 *b = bits + first_bit / 8;
    u64 t = 0;
    for (; t + 8 <= nbits; t += 8) {
        // bit k of the byte to byte k: the spread product has bit k of byte k set where bit k is set
        const u64 spread = ((u64)b[t >> 3] * 0x0101010101010101ull) & 0x8040201008040201ull;
        const u64 bytes = ((spread + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
        memcpy(dst + t, &bytes, 8);
    }
    for (; t < nbits; t++) dst[t] = (u8)((b[t >> 3] >> (t & 7)) & 1u);
}
static void pack_bits(u8 *bits, const u8 *src, u64 nbits) { // writes ceil(nbits / 8) bytes, unused bits zero
    u64 t = 0;
    for (; t + 8 <= nbits; t += 8) {
        u64 w;
        memcpy(&w, src + t, 8);
        bits[t >> 3] = (u8)(((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56); // bit 0 of byte k to bit k
    }
    if (t < nbits) {
        u8 v = 0;
        for (u64 k = t; k < nbits; k++) v |= (u8)((src[k] & 1u) << (k & 7));
        bits[t >> 3] = v;
    }
}
int mpcx_d2h_bits(mpcx_ctx *ctx, void *dst_bytes, const void *src_bits, uint64_t nbits) {
    MPCX_REQUIRE(ctx, "mpcx_d2h_bits: the context is NULL (already shut down?)");
    if (!nbits) return 0;
    MPCX_REQUIRE(dst_bytes && src_bits, "mpcx_d2h_bits: NULL pointer");
    CopyTimer timer(1, (nbits + 7) / 8);
    MPCX_TRY(stage_setup(ctx));
    const u64 packed = (nbits + 7) / 8, chunks = (packed + kStageChunk - 1) / kStageChunk;
    auto issue = [&](u64 k) -> int {
        const u64 off = k * kStageChunk, len = packed - off < kStageChunk ? packed - off : kStageChunk;
        MPCX_HIP(hipMemcpyAsync(ctx->stage[k & 1], (const char *)src_bits + off, len, hipMemcpyDeviceToHost, ctx->stream));
        MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
        return 0;
    };
    MPCX_TRY(issue(0));
    for (u64 k = 0; k < chunks; k++) {
        if (k + 1 < chunks) MPCX_TRY(issue(k + 1)); // into the other buffer, emptied in the previous iteration
        MPCX_HIP(hipEventSynchronize(ctx->stage_ev[k & 1]));
        const u64 bit0 = k * kStageChunk * 8, bits_here = nbits - bit0 < kStageChunk * 8 ? nbits - bit0 : kStageChunk * 8;
        const u8 *stage = (const u8 *)ctx->stage[k & 1];
        u8 *out = (u8 *)dst_bytes + bit0;
        const u64 piece_bits = 1ull << 20; // a megabyte of output per piece at least
        u64 pieces = bits_here / piece_bits;
        const u64 most = copy_pool().threads + 1;
        pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
        const u64 per = ((bits_here + pieces - 1) / pieces + 63) & ~63ull;
        pool_run(pieces, [=](u64 p) {
            const u64 lo = p * per;
            if (lo < bits_here) expand_bits(out + lo, stage, lo, bits_here - lo < per ? bits_here - lo : per);
        });
    }
    return 0;
}
int mpcx_h2d_bits(mpcx_ctx *ctx, void *dst_bits, const void *src_bytes, uint64_t nbits) {
    MPCX_REQUIRE(ctx, "mpcx_h2d_bits: the context is NULL (already shut down?)");
    if (!nbits) return 0;
    MPCX_REQUIRE(dst_bits && src_bytes, "mpcx_h2d_bits: NULL pointer");
    CopyTimer timer(0, (nbits + 7) / 8);
    MPCX_TRY(stage_setup(ctx));
    // whole 32-bit words reach the device (the kernels read the string word by word), the bits beyond it zero
    const u64 packed = ((nbits + 31) / 32) * 4;
    u64 k = 0;
    for (u64 off = 0; off < packed; off += kStageChunk, k++) {
        const u64 len = packed - off < kStageChunk ? packed - off : kStageChunk;
        if (k >= 2) MPCX_HIP(hipEventSynchronize(ctx->stage_ev[k & 1])); // its previous chunk has left the buffer
        u8 *stage = (u8 *)ctx->stage[k & 1];
        const u64 bit0 = off * 8, bits_here = nbits - bit0 < len * 8 ? nbits - bit0 : len * 8;
        const u8 *in = (const u8 *)src_bytes + bit0;
        const u64 piece_bits = 1ull << 20;
        u64 pieces = bits_here / piece_bits;
        const u64 most = copy_pool().threads + 1;
        pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
        const u64 per = ((bits_here + pieces - 1) / pieces + 63) & ~63ull;
        pool_run(pieces, [=](u64 p) {
            const u64 lo = p * per;
            if (lo < bits_here) pack_bits(stage + lo / 8, in + lo, bits_here - lo < per ? bits_here - lo : per);
        });
        for (u64 b = (bits_here + 7) / 8; b < len; b++) stage[b] = 0; // the rest of the last word
        MPCX_HIP(hipMemcpyAsync((char *)dst_bits + off, stage, len, hipMemcpyHostToDevice, ctx->stream));
        MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
    }
    MPCX_HIP(hipStreamSynchronize(ctx->stream)); // the staging buffers are free again when the call returns
    return 0;
}
} // extern "C"
// ---- 16-bit payloads whose values fit a byte.  The Z_67 shares of privateCompare are uint16 on the wire
// (SecurennEngine.cpp:559-589) and never exceed 66: such arrays cross the link as bytes -- narrowed by a kernel and
// widened by the copy threads on the way down, narrowed by the copy threads and widened by a kernel on the way up.
// Nothing is assumed: the narrowing pass reports a value above 255, and the array then goes as it is.
namespace {
__device__ inline u32 narrow2(u32 w) { return (w & 0xFFu) | ((w >> 8) & 0xFF00u); } // two uint16 -> two bytes
__global__ __launch_bounds__(256) void narrow_u16_kernel(const u16 *__restrict__ src, u64 count, u8 *__restrict__ dst,
                                                         u32 *__restrict__ wide, int vec) {
    u32 high = 0;
    if (vec) { // src 16-byte aligned: eight values per lane
        const u64 groups = count / 8;
        for (u64 g = (u64)blockIdx.x * 256 + threadIdx.x; g < groups; g += (u64)gridDim.x * 256) {
            const uint4 v = ((const uint4 *)src)[g];
            high |= (v.x | v.y | v.z | v.w) & 0xFF00FF00u;
            ((uint2 *)dst)[g] = make_uint2(narrow2(v.x) | narrow2(v.y) << 16, narrow2(v.z) | narrow2(v.w) << 16);
        }
        for (u64 i = groups * 8 + (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) {
            high |= src[i] & 0xFF00u;
            dst[i] = (u8)src[i];
        }
    } else {
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) {
            high |= src[i] & 0xFF00u;
            dst[i] = (u8)src[i];
        }
    }
    if (high) atomicOr(wide, 1u);
}
__global__ __launch_bounds__(256) void widen_u8_kernel(const u8 *__restrict__ src, u64 count, u16 *__restrict__ dst, int vec) {
    if (vec) { // dst 16-byte aligned
        const u64 groups = count / 8;
        for (u64 g = (u64)blockIdx.x * 256 + threadIdx.x; g < groups; g += (u64)gridDim.x * 256) {
            const uint2 v = ((const uint2 *)src)[g];
            auto w = [](u32 b) { return (b & 0xFFu) | ((b & 0xFF00u) << 8); };
            ((uint4 *)dst)[g] = make_uint4(w(v.x), w(v.x >> 16), w(v.y), w(v.y >> 16));
        }
        for (u64 i = groups * 8 + (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) dst[i] = src[i];
    } else {
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) dst[i] = src[i];
    }
}
bool narrow_u16_payloads() { // MPCX_U16_WIRE=wide: 16-bit arrays always cross the link as they are (comparison)
    static const bool on = [] {
        const char *e = getenv("MPCX_U16_WIRE");
        return !(e && strcmp(e, "wide") == 0);
    }();
    return on;
}
u32 grid_1d(u64 items) {
    u64 b = (items + 255) / 256;
    return (u32)(b < 1 ? 1 : b > 65536 ? 65536 : b);
}
// host side, on possibly unaligned arrays
void widen_host(u8 *dst16, const u8 *src8, u64 count) {
    for (u64 i = 0; i < count; i++) {
        const u16 v = src8[i];
        memcpy(dst16 + 2 * i, &v, 2);
    }
}
u32 narrow_host(u8 *dst8, const u8 *src16, u64 count) { // returns the OR of the high bytes
    u32 high = 0;
    for (u64 i = 0; i < count; i++) {
        u16 v;
        memcpy(&v, src16 + 2 * i, 2);
        high |= v & 0xFF00u;
        dst8[i] = (u8)v;
    }
    return high;
}
const u64 kNarrowHeader = 256; // the device buffer starts with the flag word, the bytes follow 256 bytes in
} // namespace
extern "C" {
int mpcx_d2h_u16(mpcx_ctx *ctx, void *dst, const void *src, uint64_t count) {
    MPCX_REQUIRE(ctx, "mpcx_d2h_u16: the context is NULL (already shut down?)");
    if (!count) return 0;
    MPCX_REQUIRE(dst && src, "mpcx_d2h_u16: NULL pointer");
    if (2 * count < kStageMin || !staged_copies() || !narrow_u16_payloads()) return mpcx_d2h(ctx, dst, src, 2 * count);
    MPCX_TRY(stage_setup(ctx));
    void *tmp = nullptr;
    MPCX_TRY(mpcx_malloc(ctx, kNarrowHeader + count, &tmp));
    auto fail = [&](int rc) {
        (void)mpcx_free(ctx, tmp);
        return rc;
    };
    if (hipMemsetAsync(tmp, 0, 4, ctx->stream) != hipSuccess) return fail(1);
    hipLaunchKernelGGL(narrow_u16_kernel, dim3(grid_1d((count + 7) / 8)), dim3(256), 0, ctx->stream, (const u16 *)src, count,
                       (u8 *)tmp + kNarrowHeader, (u32 *)tmp, ((uintptr_t)src & 15) == 0 ? 1 : 0);
    CopyTimer timer(1, count);
    const u64 total = kNarrowHeader + count, chunks = (total + kStageChunk - 1) / kStageChunk;
    auto issue = [&](u64 k) -> int {
        const u64 off = k * kStageChunk, len = total - off < kStageChunk ? total - off : kStageChunk;
        MPCX_HIP(hipMemcpyAsync(ctx->stage[k & 1], (const char *)tmp + off, len, hipMemcpyDeviceToHost, ctx->stream));
        MPCX_HIP(hipEventRecord(ctx->stage_ev[k & 1], ctx->stream));
        return 0;
    };
    if (issue(0)) return fail(1);
    for (u64 k = 0; k < chunks; k++) {
        if (k + 1 < chunks && issue(k + 1)) return fail(1);
        if (hipEventSynchronize(ctx->stage_ev[k & 1]) != hipSuccess) return fail(1);
        if (k == 0 && *(const u32 *)ctx->stage[0] != 0) { // a value above 255: the array goes as it is
            (void)hipStreamSynchronize(ctx->stream);       // (the next chunk may be on its way into the other buffer)
            (void)mpcx_free(ctx, tmp);
            return mpcx_d2h(ctx, dst, src, 2 * count);
        }
        const u64 off = k * kStageChunk, len = total - off < kStageChunk ? total - off : kStageChunk;
        const u64 skip = k == 0 ? kNarrowHeader : 0;          // header bytes at the front of this chunk
        const u64 first = off + skip - kNarrowHeader, here = len - skip; // values [first, first + here)
        const u8 *stage = (const u8 *)ctx->stage[k & 1] + skip;
        u8 *out = (u8 *)dst + 2 * first;
        u64 pieces = here >> 19; // half a megabyte of bytes (a megabyte of output) per piece at least
        const u64 most = copy_pool().threads + 1;
        pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
        const u64 per = (here + pieces - 1) / pieces;
        pool_run(pieces, [=](u64 p) {
            const u64 lo = p * per;
            if (lo < here) widen_host(out + 2 * lo, stage + lo, here - lo < per ? here - lo : per);
        });
    }
    return fail(0);
}
int mpcx_h2d_u16(mpcx_ctx *ctx, void *dst, const void *src, uint64_t count) {
    MPCX_REQUIRE(ctx, "mpcx_h2d_u16: the context is NULL (already shut down?)");
    if (!count) return 0;
    MPCX_REQUIRE(dst && src, "mpcx_h2d_u16: NULL pointer");
    if (2 * count < kStageMin || !staged_copies() || !narrow_u16_payloads()) return mpcx_h2d(ctx, dst, src, 2 * count);
    MPCX_TRY(stage_setup(ctx));
    void *tmp = nullptr;
    MPCX_TRY(mpcx_malloc(ctx, count, &tmp));
    CopyTimer timer(0, count);
    auto high = std::make_shared<std::atomic<u32>>(0u);
    u64 k = 0;
    for (u64 off = 0; off < count; off += kStageChunk, k++) {
        const u64 len = count - off < kStageChunk ? count - off : kStageChunk;
        if (k >= 2 && hipEventSynchronize(ctx->stage_ev[k & 1]) != hipSuccess) break; // its previous chunk has left the buffer
        u8 *stage = (u8 *)ctx->stage[k & 1];
        const u8 *in = (const u8 *)src + 2 * off;
        u64 pieces = len >> 19;
        const u64 most = copy_pool().threads + 1;
        pieces = pieces < 1 ? 1 : pieces > most ? most : pieces;
        const u64 per = (len + pieces - 1) / pieces;
        pool_run(pieces, [=](u64 p) {
            const u64 lo = p * per;
            if (lo < len) high->fetch_or(narrow_host(stage + lo, in + 2 * lo, len - lo < per ? len - lo : per));
        });
        if (high->load()) break;
        if (hipMemcpyAsync((char *)tmp + off, stage, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ctx->stage_ev[k & 1], ctx->stream) != hipSuccess) {
            high->store(0xFFFFFFFFu); // treated like a wide value: the plain path reports the error if it persists
            break;
        }
    }
    if (high->load()) { // a value above 255 (or a copy that would not start): the array goes as it is
        (void)hipStreamSynchronize(ctx->stream);
        (void)mpcx_free(ctx, tmp);
        return mpcx_h2d(ctx, dst, src, 2 * count);
    }
    hipLaunchKernelGGL(widen_u8_kernel, dim3(grid_1d((count + 7) / 8)), dim3(256), 0, ctx->stream, (const u8 *)tmp, count,
                       (u16 *)dst, ((uintptr_t)dst & 15) == 0 ? 1 : 0);
    const hipError_t e = hipStreamSynchronize(ctx->stream); // the staging buffers and tmp are free again
    (void)mpcx_free(ctx, tmp);
    MPCX_HIP(e);
    return 0;
}
int mpcx_host_copy(void *dst, const void *src, uint64_t bytes) {
    CopyTimer timer(2, bytes);
    if (bytes) mpcx_parallel_memcpy(dst, src, bytes);
    return 0;
}
int mpcx_host_transpose64(void *dst, const void *src, uint64_t rows, uint64_t cols) {
    if (!rows || !cols) return 0;
    if (!dst || !src || dst == src) return 1;
    CopyTimer timer(2, 8 * rows * cols);
    CopyPool &P = copy_pool();
    u64 pieces = rows * cols < (1ull << 16) ? 1 : P.threads + 1;
    const u64 band = ((cols + pieces - 1) / pieces + 31) & ~31ull; // whole tiles
    pieces = (cols + band - 1) / band;
    pool_run(pieces, [=](u64 k) {
        const u64 lo = k * band, hi = lo + band < cols ? lo + band : cols;
        if (lo < hi) transpose64_band((u64 *)dst, (const u64 *)src, rows, cols, lo, hi);
    });
    return 0;
}
int mpcx_d2d(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    MPCX_REQUIRE(ctx, "mpcx_d2d: the context is NULL (already shut down?)");
    if (!bytes) return 0;
    MPCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
static const char *kProfNames[PROF_SLOTS] = {"limb_gemm", "split_limbs", "prg_fill", "pc_party", "pc_helper", "msb_round"};

int mpcx_profile(mpcx_ctx *ctx, int enable) {
    MPCX_REQUIRE(ctx, "mpcx_profile: the context is NULL (already shut down?)");
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &v : ctx->prof_ev) {
        for (auto &p : v) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        v.clear();
    }
    ctx->prof = enable != 0;
    return 0;
}
int mpcx_profile_select(mpcx_ctx *ctx, const char *kernel) {
    MPCX_REQUIRE(ctx, "mpcx_profile_select: the context is NULL (already shut down?)");
    if (!kernel) {
        ctx->prof_mask = ~0u;
        return 0;
    }
    for (int s = 0; s < PROF_SLOTS; s++)
        if (std::string(kernel) == kProfNames[s]) {
            ctx->prof_mask = 1u << s;
            return 0;
        }
    return mpcx_fail(ctx, "mpcx_profile_select: unknown kernel name '%s'", kernel);
}
int mpcx_profile_read(mpcx_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches) {
    MPCX_REQUIRE(ctx, "mpcx_profile_read: the context is NULL (already shut down?)");
    MPCX_REQUIRE(kernel && total_ms && launches, "mpcx_profile_read: NULL pointer");
    MPCX_HIP(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < PROF_SLOTS; s++) {
        if (std::string(kernel) != kProfNames[s]) continue;
        double tot = 0;
        for (auto &p : ctx->prof_ev[s]) {
            float ms = 0;
            MPCX_HIP(hipEventElapsedTime(&ms, p.first, p.second));
            tot += ms;
        }
        *total_ms = tot;
        *launches = ctx->prof_ev[s].size();
        return 0;
    }
    return mpcx_fail(ctx, "mpcx_profile_read: unknown kernel name '%s'", kernel);
}

int mpcx_reserve_workspace(mpcx_ctx *ctx, uint64_t bytes) {
    MPCX_REQUIRE(ctx, "mpcx_reserve_workspace: the context is NULL (already shut down?)");
    void *p;
    return mpcx_ws(ctx, bytes, &p);
}

} // extern "C"
