// On-device regeneration of the reference PRgenerator streams (utils/PRgenerator.cpp:3-42).
//
// prvhash-core64 has no jump-ahead (the state advances through a data-dependent multiply), so the
// stream is made parallel with a checkpoint table: the host walks the stream once and keeps the
// 24-byte state every PRG_CK outputs; one GPU lane then replays one PRG_CK-word segment.  A
// 256-thread workgroup therefore owns 256 consecutive segments; every 16 steps the lanes
// park their words in LDS and the workgroup writes them out as whole 128-byte lines.
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mpcx_internal.h"
#include "prg_map.h"

namespace {

// The host copy of a checkpoint table: a contiguous array that grows by remapping its pages (mremap) instead of
// copying them.  A table reaches gigabytes (0.375 bytes per stream word), it grows while engine calls wait for its lock,
// and a std::vector would copy all of it whenever it outgrows its capacity -- hundreds of milliseconds in front of
// whichever call comes next.  Fresh pages are zero; entries are plain data.
class CkVec {
    PrgCk *p_ = nullptr;
    u64 n_ = 0, cap_ = 0;
    static u64 bytes_for(u64 cap) { return (cap * sizeof(PrgCk) + (2u << 20) - 1) & ~(u64)((2u << 20) - 1); }

  public:
    CkVec() = default;
    CkVec(const CkVec &) = delete;
    CkVec &operator=(const CkVec &) = delete;
    ~CkVec() {
        if (p_) munmap(p_, bytes_for(cap_));
    }
    u64 size() const { return n_; }
    u64 capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    PrgCk *data() { return p_; }
    const PrgCk *data() const { return p_; }
    PrgCk &operator[](u64 i) { return p_[i]; }
    const PrgCk &operator[](u64 i) const { return p_[i]; }
    const PrgCk &back() const { return p_[n_ - 1]; }
    void reserve(u64 cap) { // throws std::bad_alloc like the container it replaces
        if (cap <= cap_) return;
        const u64 nb = bytes_for(cap);
        void *q = p_ ? mremap(p_, bytes_for(cap_), nb, MREMAP_MAYMOVE)
                     : mmap(nullptr, nb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        p_ = (PrgCk *)q;
        cap_ = nb / sizeof(PrgCk);
    }
    void grow_for(u64 want) { // room for `want` entries, at least half as much again as there is
        if (want <= cap_) return;
        const u64 geo = cap_ + cap_ / 2;
        reserve(want > geo ? want : geo);
    }
    void resize(u64 n) { // new entries are unspecified (zero when the pages are fresh)
        grow_for(n);
        n_ = n;
    }
    void push_back(const PrgCk &c) {
        grow_for(n_ + 1);
        p_[n_++] = c;
    }
    void append(const PrgCk *src, u64 count) {
        grow_for(n_ + count);
        memcpy(p_ + n_, src, count * sizeof(PrgCk));
        n_ += count;
    }
    void swap(CkVec &o) {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        std::swap(cap_, o.cap_);
    }
};

struct Table {
    std::mutex mu; // walk and upload of THIS stream; streams are extended independently of one another
    std::atomic<int> waiters{0}; // engine calls waiting for mu: the background thread stays away while there are any
    CkVec host; // host[e] = state before stream word PRG_CK*e
    PrgCk *dev = nullptr;
    u64 dev_cap = 0, dev_count = 0;
    // Older, smaller device copies.  A table is shared by every context of the device, so a kernel of another
    // thread may still read (or be about to be launched on) a copy when the table grows: copies are kept until the
    // last context of the device shuts down.  Growth is geometric, so together they hold less than the live copy.
    std::vector<PrgCk *> retired;
    u64 bg_target = 0;    // checkpoints the background walker has been asked for
    bool job_queued = false; // a job for this table is in the walker's queue or running
    u64 last_need = 0;    // the previous request, for the rate at which the stream is consumed
    u64 loaded_from_disk = 0, saved_to_disk = 0;
    u64 disk_seen_size = 0, disk_seen_mtime = 0; // the cache file as ckpt_load last saw it
};
// An engine call's hold on a table.
struct CallerLock {
    Table &T;
    explicit CallerLock(Table &t) : T(t) {
        T.waiters.fetch_add(1);
        T.mu.lock();
        T.waiters.fetch_sub(1);
    }
    ~CallerLock() { T.mu.unlock(); }
    CallerLock(const CallerLock &) = delete;
    CallerLock &operator=(const CallerLock &) = delete;
};
// The background thread's hold: short (one slice of a walk appended, one chunk of an upload or of a file written), and
// given up at once when an engine call has asked for the table in the meantime -- a plain mutex would let the thread
// that releases and re-takes it in a loop win against a waiting caller again and again.
struct WalkerLock {
    Table &T;
    explicit WalkerLock(Table &t) : T(t) {
        for (;;) {
            while (T.waiters.load() > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
            T.mu.lock();
            if (T.waiters.load() == 0) return;
            T.mu.unlock();
        }
    }
    ~WalkerLock() { T.mu.unlock(); }
    WalkerLock(const WalkerLock &) = delete;
    WalkerLock &operator=(const WalkerLock &) = delete;
};
struct Key {
    int device;
    u64 seed, lcg;
    bool operator<(const Key &o) const {
        if (device != o.device) return device < o.device;
        if (seed != o.seed) return seed < o.seed;
        return lcg < o.lcg;
    }
};
// The registry and the walker's queue are never destroyed: the background thread may still run at process exit.
struct Registry {
    std::mutex mu; // guards the map itself (its nodes never move)
    std::map<Key, Table> tables;
    std::map<int, int> live_contexts; // per device
    std::mutex smu; // guards upload_streams only; nothing else is taken while it is held
    std::map<int, hipStream_t> upload_streams; // per device, never destroyed
    // background walker: one thread, jobs = (table, device)
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<std::pair<Table *, Key>> queue;
    bool walker_started = false, stop = false, walker_busy = false;
};
Registry &reg() {
    static Registry *r = new Registry();
    return *r;
}
Table &table_of(int device, u64 seed, u64 lcg) {
    Registry &R = reg();
    std::lock_guard<std::mutex> g(R.mu);
    return R.tables[Key{device, seed, lcg}];
}

// ---- checkpoints kept on disk.  The keys of the engines are compile-time constants (mpc_engine/ABY3Engine.hpp:10-15),
// so the serial walk of a stream is the same work in every process and on every rank: a table is read from
// <dir>/prg_<seed>_<lcg>.ckpt before anything is walked, and written back (to a temporary name, then renamed) when it
// has grown by more than a quarter.  <dir> = $MPCX_CKPT_CACHE when set (empty: no cache), else $XDG_CACHE_HOME/mpcx,
// else $HOME/.cache/mpcx.  File = 40-byte header {magic, seed, lcg, count, checksum of the body} followed by count
// checkpoints.  A file is trusted only when its length is the one its header implies, the checksum matches, its first
// entry is the state this process computed itself and its last entry follows from the one before by PRG_CK steps;
// anything else is ignored (and later overwritten).
const u64 CKPT_MAGIC = 0x3243504B435850ull + ((u64)PRG_CK << 56);
const u64 CKPT_HDR_BYTES = 40;
std::string ckpt_path(u64 seed, u64 lcg) {
    std::string dir;
    const char *env = getenv("MPCX_CKPT_CACHE");
    if (env) {
        if (!*env) return std::string();
        dir = env;
    } else {
        const char *xdg = getenv("XDG_CACHE_HOME"), *home = getenv("HOME");
        if (xdg && *xdg)
            dir = std::string(xdg);
        else if (home && *home)
            dir = std::string(home) + "/.cache";
        else
            return std::string();
        (void)mkdir(dir.c_str(), 0700); // the parent, then the library's own directory; both may exist
        dir += "/mpcx";
        (void)mkdir(dir.c_str(), 0700);
    }
    char name[96];
    snprintf(name, sizeof name, "/prg_%016llx_%016llx.ckpt", (unsigned long long)seed, (unsigned long long)lcg);
    return dir + name;
}
// four interleaved multiply-xor lanes over the 64-bit words of the body (a serial chain would cost as much as the walk);
// fed in pieces, every piece but the last a multiple of four checkpoints
struct CkptSum {
    u64 h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    u64 nw = 0;
    void add(const PrgCk *v, u64 count) {
        const u64 *w = (const u64 *)v;
        const u64 n = count * (sizeof(PrgCk) / 8);
        u64 i = 0;
        for (; i + 4 <= n; i += 4)
            for (int k = 0; k < 4; k++) {
                h[k] = (h[k] ^ w[i + k]) * 0x9E3779B97F4A7C15ull;
                h[k] ^= h[k] >> 29;
            }
        for (; i < n; i++) h[0] = ((h[0] ^ w[i]) * 0x9E3779B97F4A7C15ull) ^ (h[0] >> 31);
        nw += n;
    }
    u64 value() const { return (h[0] * 3 + h[1] * 5 + h[2] * 7 + h[3] * 11) ^ nw; }
};
u64 ckpt_checksum(const PrgCk *v, u64 count) {
    CkptSum s;
    s.add(v, count);
    return s.value();
}
void ckpt_load(Table &T, u64 seed, u64 lcg) {
    const std::string path = ckpt_path(seed, lcg);
    if (path.empty() || T.host.empty()) return;
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return;
    // the file this table last looked at (same length and modification time): nothing new in it
    if ((u64)st.st_size == T.disk_seen_size && (u64)st.st_mtime == T.disk_seen_mtime) return;
    T.disk_seen_size = (u64)st.st_size;
    T.disk_seen_mtime = (u64)st.st_mtime;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return;
    u64 hdr[5];
    const bool hdr_ok = fread(hdr, 8, 5, f) == 5 && hdr[0] == CKPT_MAGIC && hdr[1] == seed && hdr[2] == lcg &&
                        hdr[3] > T.host.size() && hdr[3] < (1ull << 36) &&
                        (u64)st.st_size == CKPT_HDR_BYTES + hdr[3] * sizeof(PrgCk);
    if (hdr_ok) {
        try {
            CkVec v;
            v.resize(hdr[3]);
            // a file that another process is replacing is complete or absent (rename); a short read is ignored
            bool ok = fread(v.data(), sizeof(PrgCk), hdr[3], f) == hdr[3] && v[0].seed == T.host[0].seed &&
                      v[0].lcg == T.host[0].lcg && v[0].hash == T.host[0].hash && ckpt_checksum(v.data(), hdr[3]) == hdr[4];
            if (ok && hdr[3] >= 2) {
                PrgCk c = v[hdr[3] - 2];
                for (u64 i = 0; i < PRG_CK; i++) prv_step(c.seed, c.lcg, c.hash);
                const PrgCk &l = v[hdr[3] - 1];
                ok = c.seed == l.seed && c.lcg == l.lcg && c.hash == l.hash;
            }
            if (ok) {
                T.host.swap(v);
                T.loaded_from_disk = T.saved_to_disk = T.host.size();
            }
        } catch (const std::exception &) { // no memory for a table of that size: walk instead
        }
    }
    fclose(f);
}
const u64 CKPT_CHUNK = 262144; // checkpoints (6 MB) the background thread handles per hold of a table's lock
// Writes the table to the cache.  background = false: the caller holds T.mu throughout.  background = true (the
// background thread, T.mu not held): the lock is taken for one chunk at a time, so an engine call that needs the table
// waits for a few megabytes at most; the entries being written never change (a table only grows), and `stopping`
// abandons the write.
template <class Stop> void ckpt_save(Table &T, u64 seed, u64 lcg, bool background, Stop stopping) {
    const std::string path = ckpt_path(seed, lcg);
    if (path.empty()) return;
    auto locked = [&](auto &&f) {
        if (background) {
            WalkerLock g(T);
            f();
        } else {
            f();
        }
    };
    u64 count = 0;
    locked([&] {
        count = T.host.size();
        if (count <= T.saved_to_disk + T.saved_to_disk / 4 + 4096) count = 0;
    });
    // in the implicit directory (nobody asked for a cache) tables beyond 256 MiB stay in memory only: with the sparse
    // marks a long stream is walked about as fast as such a file is read, and a home directory is no place for gigabytes
    if (!count || (!getenv("MPCX_CKPT_CACHE") && count * sizeof(PrgCk) > (256ull << 20))) return;
    char tmp[32];
    snprintf(tmp, sizeof tmp, ".tmp.%ld", (long)getpid());
    const std::string tpath = path + tmp;
    FILE *f = fopen(tpath.c_str(), "wb");
    if (!f) return;
    u64 hdr[5] = {0, 0, 0, 0, 0}; // the real header goes in when the body is complete
    bool ok = fwrite(hdr, 8, 5, f) == 5;
    CkptSum sum;
    for (u64 done = 0; ok && done < count;) {
        const u64 n = count - done < CKPT_CHUNK ? count - done : CKPT_CHUNK;
        locked([&] {
            const PrgCk *src = T.host.data() + done;
            ok = fwrite(src, sizeof(PrgCk), n, f) == n;
            sum.add(src, n);
        });
        done += n;
        if (background && stopping()) ok = false;
    }
    hdr[0] = CKPT_MAGIC, hdr[1] = seed, hdr[2] = lcg, hdr[3] = count, hdr[4] = sum.value();
    ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(hdr, 8, 5, f) == 5;
    if (fclose(f) == 0 && ok && rename(tpath.c_str(), path.c_str()) == 0)
        locked([&] {
            if (T.saved_to_disk < count) T.saved_to_disk = count;
        });
    else
        remove(tpath.c_str());
}
void ckpt_save(Table &T, u64 seed, u64 lcg) { // T.mu held
    ckpt_save(T, seed, lcg, false, [] { return false; });
}

// ---- sparse marks: the state of a stream every MARK_STRIDE words, for the streams whose keys are compile-time
// constants of the engines (and the benchmark's input streams), shipped beside the library as prg_marks.bin
// (tools/make_prg_marks.cpp; MPCX_PRG_MARKS=<file> overrides, an empty value switches the marks off).  prvhash-core64
// has no jump-ahead, so one stream is one serial chain (~1 ns per word on a host core: 5 s for the 2^32 words a
// 2^24-element privateCompare reads); with the marks, the segments between them are walked on many threads at once.
// Nothing in the file is trusted: mark 0 must be the state this process computes itself, and a segment walked from mark k
// must end in mark k + 1 -- otherwise the parallel result is discarded and the serial walk takes over.
struct Marks {
    u64 stride = 0;
    std::map<std::pair<u64, u64>, std::vector<PrgCk>> by_key;
};
const Marks &marks() {
    static Marks M;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string path;
        const char *env = getenv("MPCX_PRG_MARKS");
        if (env) {
            path = env;
        } else {
            Dl_info info;
            if (dladdr((const void *)&marks, &info) && info.dli_fname) {
                path = info.dli_fname;
                const size_t slash = path.rfind('/');
                path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/prg_marks.bin";
            }
        }
        if (path.empty()) return;
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return;
        u64 hdr[3];
        if (fread(hdr, 8, 3, f) == 3 && hdr[0] == 0x314B524D5843504Dull && hdr[1] >= PRG_CK && hdr[1] % PRG_CK == 0 &&
            hdr[2] <= 64) {
            Marks tmp;
            tmp.stride = hdr[1];
            bool ok = true;
            for (u64 k = 0; k < hdr[2] && ok; k++) {
                u64 kh[3];
                ok = fread(kh, 8, 3, f) == 3 && kh[2] >= 1 && kh[2] <= (1u << 20);
                if (!ok) break;
                std::vector<PrgCk> v(kh[2]);
                ok = fread(v.data(), sizeof(PrgCk), kh[2], f) == kh[2];
                if (ok) tmp.by_key[std::make_pair(kh[0], kh[1])].swap(v);
            }
            if (ok) M = std::move(tmp);
        }
        fclose(f);
    });
    return M;
}

// extends T.host to at least `want` checkpoints, one after the other (T.mu held)
void walk_serial(Table &T, u64 want) {
    PrgCk c = T.host.back();
    while (T.host.size() < want) {
        for (u64 i = 0; i < PRG_CK; i++) prv_step(c.seed, c.lcg, c.hash);
        T.host.push_back(c);
    }
}
// the same on several threads, segment by segment between the marks of the stream; false = nothing was changed
bool walk_with_marks(Table &T, u64 seed, u64 lcg, u64 want) {
    const Marks &M = marks();
    if (!M.stride) return false;
    const auto it = M.by_key.find(std::make_pair(seed, lcg));
    if (it == M.by_key.end()) return false;
    const std::vector<PrgCk> &mk = it->second;
    const PrgCk &c0 = T.host[0];
    if (mk[0].seed != c0.seed || mk[0].lcg != c0.lcg || mk[0].hash != c0.hash) return false;
    const u64 per = M.stride / PRG_CK;                 // checkpoints per segment
    const u64 have = T.host.size();                    // checkpoints 0 .. have - 1 exist
    const u64 first = (have - 1 + per - 1) / per;      // first mark at or after the last checkpoint there is
    u64 last = (want - 1) / per;                       // last mark at or before the last checkpoint wanted
    if (last >= mk.size()) last = mk.size() - 1;
    if (last <= first || last - first < 2) return false; // too little to share out
    // the stretch before the first usable mark, serially (less than one segment)
    walk_serial(T, first * per + 1);
    const PrgCk &at = T.host[first * per];
    if (at.seed != mk[first].seed || at.lcg != mk[first].lcg || at.hash != mk[first].hash) return false; // walked anyway
    const u64 old_size = T.host.size();
    T.host.resize(last * per + 1);
    const u64 nseg = last - first;
    unsigned nth = std::thread::hardware_concurrency();
    nth = nth < 2 ? 2 : nth > 16 ? 16 : nth;
    if (nth > nseg) nth = (unsigned)nseg;
    std::atomic<u64> next{0};
    std::atomic<bool> good{true};
    auto work = [&] {
        for (;;) {
            const u64 k = next.fetch_add(1);
            if (k >= nseg || !good.load()) return;
            const u64 m = first + k;
            PrgCk c = mk[m];
            PrgCk *out = T.host.data() + m * per; // out[0] is mark m itself
            out[0] = c;
            for (u64 e = 1; e <= per; e++) {
                for (u64 i = 0; i < PRG_CK; i++) prv_step(c.seed, c.lcg, c.hash);
                if (e < per) out[e] = c;
            }
            if (c.seed != mk[m + 1].seed || c.lcg != mk[m + 1].lcg || c.hash != mk[m + 1].hash) good.store(false);
        }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nth; i++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (!good.load()) { // a mark does not follow from the one before it: keep only what was walked serially
        T.host.resize(old_size);
        return false;
    }
    T.host[last * per] = mk[last];
    return true;
}
// extends T.host to at least `want` checkpoints (T.mu held)
void walk_to(Table &T, u64 want, u64 seed = 0, u64 lcg = 0, bool keyed = false) {
    if (T.host.size() >= want) return;
    T.host.grow_for(want);
    if (keyed && want - T.host.size() >= (1u << 17)) (void)walk_with_marks(T, seed, lcg, want); // 2^23 words and more
    walk_serial(T, want);
}
// Table uploads run on a stream of their own that does not synchronise with the null stream: an engine that works on
// the null stream (torch's default) has kernels queued there, and a plain hipMemcpy would wait for all of them.
hipError_t upload_stream(int device, hipStream_t *out) { // the device current
    Registry &R = reg();
    std::lock_guard<std::mutex> g(R.smu);
    auto it = R.upload_streams.find(device);
    if (it == R.upload_streams.end()) {
        hipStream_t s = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        it = R.upload_streams.emplace(device, s).first;
    }
    *out = it->second;
    return hipSuccess;
}
hipError_t upload_copy(PrgCk *dst, const PrgCk *src, u64 count, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(dst, src, count * sizeof(PrgCk), hipMemcpyHostToDevice, s);
    return e == hipSuccess ? hipStreamSynchronize(s) : e; // complete before anybody is told that the entries exist
}
// room of a device copy for `have` checkpoints: the first copy gets a sixteenth more than the request (a table built to a
// stated length is often read a few words further); a table that outgrows a copy is a pop stream on the move and gets
// half as much spare room again, so that the superseded copies stay a geometric series
u64 upload_capacity(const Table &T, u64 have) {
    const u64 room = T.dev ? have / 2 : have / 16;
    return have + (room > 4096 ? room : 4096);
}
// mirrors T.host on the device (T.mu held, the device current); returns a hipError_t
hipError_t upload(Table &T, int device) {
    const u64 have = T.host.size();
    if (T.dev_count >= have) return hipSuccess;
    hipStream_t s;
    hipError_t e = upload_stream(device, &s);
    if (e != hipSuccess) return e;
    if (T.dev_cap < have) {
        const u64 cap = upload_capacity(T, have);
        PrgCk *nd = nullptr;
        e = hipMalloc((void **)&nd, cap * sizeof(PrgCk));
        if (e != hipSuccess) return e;
        if (T.dev) T.retired.push_back(T.dev);
        T.dev = nd;
        T.dev_cap = cap;
        T.dev_count = 0;
    }
    e = upload_copy(T.dev + T.dev_count, T.host.data() + T.dev_count, have - T.dev_count, s);
    if (e == hipSuccess) T.dev_count = have;
    return e;
}
// The same from the background thread (T.mu NOT held, the device current): the lock is taken for one chunk at a time,
// and a larger device copy is filled completely before the table points at it, so an engine call never waits for more
// than a few megabytes and never sees a copy that is half there.  false = given up (error, or the process is ending).
template <class Stop> bool upload_background(Table &T, int device, Stop stopping) {
    hipStream_t s;
    if (upload_stream(device, &s) != hipSuccess) return false;
    PrgCk *fresh = nullptr; // a larger copy in the making
    u64 fresh_cap = 0, filled = 0;
    for (;;) {
        if (stopping()) break;
        u64 want_cap = 0;
        {
            WalkerLock g(T);
            const u64 have = T.host.size();
            if (T.dev_count >= have) { // the live copy is complete (an engine call may have seen to it)
                if (fresh) (void)hipFree(fresh);
                return true;
            }
            if (fresh && T.dev_cap >= have && T.dev_count >= filled) {
                // an engine call has installed a copy with room of its own meanwhile: this one is superfluous
                (void)hipFree(fresh);
                fresh = nullptr;
            }
            if (!fresh && T.dev_cap >= have) { // room in the live copy: append a chunk
                const u64 n = have - T.dev_count < CKPT_CHUNK ? have - T.dev_count : CKPT_CHUNK;
                if (upload_copy(T.dev + T.dev_count, T.host.data() + T.dev_count, n, s) != hipSuccess) return false;
                T.dev_count += n;
                continue;
            }
            if (fresh) {
                const u64 upto = have < fresh_cap ? have : fresh_cap;
                const u64 n = upto - filled < CKPT_CHUNK ? upto - filled : CKPT_CHUNK;
                if (n && upload_copy(fresh + filled, T.host.data() + filled, n, s) != hipSuccess) break;
                filled += n;
                if (filled == upto) { // caught up with the table: the new copy takes over
                    if (filled > T.dev_count) {
                        if (T.dev) T.retired.push_back(T.dev);
                        T.dev = fresh;
                        T.dev_cap = fresh_cap;
                        T.dev_count = filled;
                    } else {
                        (void)hipFree(fresh);
                    }
                    fresh = nullptr;
                }
                continue;
            }
            want_cap = upload_capacity(T, have);
        }
        // a larger copy is needed: allocated without the lock
        if (hipMalloc((void **)&fresh, want_cap * sizeof(PrgCk)) != hipSuccess) return false;
        fresh_cap = want_cap;
        filled = 0;
    }
    if (fresh) (void)hipFree(fresh);
    return false;
}

// The background walker: a pop stream is consumed a little further by every engine call, and its serial walk
// (~1 ns per word on one core) would otherwise sit in front of every call.  After a request has been served, the
// walker carries the table on to bg_target: it walks a slice from the table's last state WITHOUT the table's lock and
// appends it under the lock (unless an engine call has extended the table itself in the meantime), then mirrors the table
// on the device and writes it to the cache, a chunk per hold of the lock.
void walker_main() {
    Registry &R = reg();
    std::vector<PrgCk> slice;
    for (;;) {
        std::pair<Table *, Key> job;
        {
            std::unique_lock<std::mutex> lk(R.qmu);
            R.walker_busy = false;
            R.qcv.notify_all();
            R.qcv.wait(lk, [&] { return R.stop || !R.queue.empty(); });
            if (R.stop) return;
            job = R.queue.front();
            R.queue.pop_front();
            R.walker_busy = true;
        }
        Table &T = *job.first;
        const Key key = job.second;
        auto stopping = [&] {
            std::lock_guard<std::mutex> lk(R.qmu);
            return R.stop;
        };
        // a job that is dropped (the device cannot be made current, an upload fails, or the process is ending) hands
        // the table back: with bg_target left above the table's size, walk_ahead would take the job for still running
        // and never queue another one
        auto drop = [&] {
            WalkerLock g(T);
            T.bg_target = T.host.size();
            T.job_queued = false;
        };
        if (hipSetDevice(key.device) != hipSuccess) {
            drop();
            continue;
        }
        for (bool finished = false; !finished;) {
            bool alive = true;
            for (;;) { // the walk
                if (stopping()) {
                    alive = false;
                    break;
                }
                PrgCk c;
                u64 size0, target;
                {
                    WalkerLock g(T);
                    size0 = T.host.size();
                    target = T.bg_target;
                    c = T.host.back();
                }
                if (size0 >= target) break;
                const u64 n = target - size0 < 32768 ? target - size0 : 32768; // ~2 M words, a few milliseconds
                slice.resize(n);
                for (u64 e = 0; e < n; e++) {
                    for (u64 i = 0; i < PRG_CK; i++) prv_step(c.seed, c.lcg, c.hash);
                    slice[e] = c;
                }
                WalkerLock g(T);
                if (T.host.size() != size0) continue; // an engine call walked on from the same state: its entries stand
                try {
                    T.host.grow_for(target); // pages are remapped, not copied
                    T.host.append(slice.data(), n);
                } catch (const std::bad_alloc &) { // no memory to walk ahead with: the calls walk what they need
                    alive = false;
                    break;
                }
            }
            if (!alive || !upload_background(T, key.device, stopping)) {
                drop();
                break;
            }
            ckpt_save(T, key.seed, key.lcg, true, stopping);
            WalkerLock g(T);
            if (T.host.size() >= T.bg_target) { // no further request came in meanwhile
                T.job_queued = false;
                finished = true;
            }
        }
    }
}
// at process exit (registered after the HIP runtime came up, so it runs before the runtime is torn down): the walker
// finishes its slice and stops
void walker_stop() {
    Registry &R = reg();
    std::unique_lock<std::mutex> lk(R.qmu);
    R.stop = true;
    R.qcv.notify_all();
    R.qcv.wait(lk, [&] { return !R.walker_busy; });
}
bool walker_enabled() {
    static const bool on = [] {
        const char *e = getenv("MPCX_PRG_WALK_AHEAD");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
// Asks the background thread to carry the table on to `target` checkpoints and then to mirror it on the device and
// write it to the disk cache (T.mu held).  target = 0: only the mirroring and the write-back of what has been walked.
void walk_ahead(Table &T, const Key &key, u64 target) {
    if (!walker_enabled()) return;
    if (target > T.bg_target) T.bg_target = target;
    if (T.job_queued) return; // the running job reads bg_target under T.mu and goes on to it
    if (target && target <= T.host.size()) return;
    Registry &R = reg();
    std::lock_guard<std::mutex> lk(R.qmu);
    if (R.stop) return;
    if (!R.walker_started) {
        R.walker_started = true;
        atexit(walker_stop);
        std::thread(walker_main).detach();
    }
    T.job_queued = true;
    R.queue.emplace_back(&T, key);
    R.qcv.notify_one();
}

// mode 0: every stream word -> out64[rel]                                  (pop stream)
// mode 1: even words -> out64[rel/2], odd words' low bit -> out8[rel/2]     (int_buff / bool_buff)
// mode 2: even words -> out8[rel/2] = (uint16)word % 67                     (Z_67 shares drawn from int_buff)
// mode 3: as mode 2 with 0 mapped to 1                                      (privateCompare's s and u)
__global__ __launch_bounds__(256) void prg_fill_kernel(const PrgCk *__restrict__ ck, u64 seg0, u64 nseg, u64 t_first,
                                                       u64 t_end, int mode, u64 *__restrict__ out64,
                                                       u8 *__restrict__ out8) {
    __shared__ u64 tile[256 * 17];
    const u32 tid = threadIdx.x;
    const u64 blk_seg0 = seg0 + (u64)blockIdx.x * 256;
    const u64 seg = blk_seg0 + tid;
    const bool active = seg < seg0 + nseg;
    u64 seed = 0, lcg = 0, hash = 0;
    if (active) {
        PrgCk c = ck[seg];
        seed = c.seed;
        lcg = c.lcg;
        hash = c.hash;
    }
    for (u32 it = 0; it < PRG_CK / 16; it++) {
#pragma unroll
        for (u32 k = 0; k < 16; k++) tile[tid * 17 + k] = prv_step(seed, lcg, hash);
        prg_wave_sync(); // the exchange stays inside the wavefront (prg_map.h)
#pragma unroll 4
        for (u32 j = 0; j < 16; j++) {
            const u32 idx = j * 64 + (tid & 63);
            const u32 row = (tid & ~63u) + (idx >> 4), col = idx & 15;
            const u64 t = (blk_seg0 + row) * PRG_CK + it * 16 + col;
            if (t >= t_first && t < t_end) {
                const u64 v = tile[row * 17 + col];
                const u64 rel = t - t_first;
                if (mode == 0) {
                    out64[rel] = v;
                } else if ((rel & 1) == 0) {
                    if (mode == 1) {
                        if (out64) out64[rel >> 1] = v;
                    } else {
                        u32 m = (u32)(v & 0xFFFF) % 67u;
                        if (mode == 3 && m == 0) m = 1;
                        out8[rel >> 1] = (u8)m;
                    }
                } else if (mode == 1 && out8) {
                    out8[rel >> 1] = (u8)(v & 1);
                }
            }
        }
        prg_wave_sync();
    }
}



// out[i] = (uint16)int_buff[first + i] % 67 (0 -> 1 when zero_to_one) for a run whose first stream word is a
// checkpoint: a lane's PRG_CK stream words are PRG_CK / 2 consecutive output bytes, which it packs in registers and
// stores itself -- no exchange through LDS, no work on the discarded odd words beyond the generator step.
__global__ __launch_bounds__(256) void prg_mod67_kernel(const PrgCk *__restrict__ ck, u64 seg0, u64 count, int zero_to_one,
                                                        u8 *__restrict__ out) {
    constexpr u32 BPL = PRG_CK / 2; // bytes per lane
    const u64 lane = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 b0 = lane * BPL;
    if (b0 >= count) return;
    u32 w[BPL / 4];
    mod67_run(ck[seg0 + lane], zero_to_one, w);
    u8 *dst = out + b0; // count is a multiple of BPL (checked by the launcher)
#pragma unroll
    for (u32 q = 0; q < BPL / 16; q++) *(uint4 *)(dst + 16 * q) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// The same bytes for the listed elements only: element e owns 64 bytes = two checkpoint runs, and lane t of the
// launch fills half (t & 1) of element list[t >> 1], for t < 2 * *listed; the bytes of other elements stay untouched.
__global__ __launch_bounds__(256) void prg_mod67_listed_kernel(const PrgCk *__restrict__ ck, u64 seg0,
                                                               const u32 *__restrict__ list,
                                                               const u32 *__restrict__ listed, int zero_to_one,
                                                               u8 *__restrict__ out) {
    static_assert(PRG_CK == 64, "an element's 64 bytes are two checkpoint runs");
    const u64 total = 2 * (u64)*listed;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < total; t += (u64)gridDim.x * 256) {
        const u64 run = 2 * (u64)list[t >> 1] + (t & 1);
        u32 w[8];
        mod67_run(ck[seg0 + run], zero_to_one, w);
        u8 *dst = out + run * 32;
        *(uint4 *)dst = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

} // namespace

// save_now: write a table that grew back to the disk cache before returning (mpcx_prg_ckpt_build: another rank of
// the node may be waiting for the file); otherwise the background thread writes it, off the caller's path
static int prg_table(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_end, const PrgCk **dev, bool save_now) {
    Table &T = table_of(ctx->device, seed, lcg);
    CallerLock g(T);
    if (T.host.empty()) {
        u64 s = seed, l = lcg, h = 0;
        for (int i = 0; i < 5; i++) prv_step(s, l, h); // utils/PRgenerator.cpp:23-26
        T.host.push_back(PrgCk{s, l, h});
        ckpt_load(T, seed, lcg);
    }
    const u64 need = t_end / PRG_CK + 1; // checkpoints 0 .. floor(t_end/PRG_CK)
    if (T.host.size() < need) ckpt_load(T, seed, lcg); // another process of the node may have walked further already
    if (T.host.size() < need) {
        // walk a little past the request (between 4 Ki and 64 Ki checkpoints) so that the device copy grows once in
        // many calls
        u64 ahead = need / 8;
        ahead = ahead < 4096 ? 4096 : ahead > 65536 ? 65536 : ahead;
        walk_to(T, need + ahead, seed, lcg, true);
        if (save_now || !walker_enabled())
            ckpt_save(T, seed, lcg);
        else
            walk_ahead(T, Key{ctx->device, seed, lcg}, 0);
    }
    if (save_now) ckpt_save(T, seed, lcg); // also what the background thread walked since the last write
    // The background walker keeps the table ahead of a stream that is being consumed: four times the step from the
    // previous request, within 16 Mi .. 256 Mi words, started when less than half of that is left.  Only a request
    // beyond everything asked before counts as consumption: the first one says nothing about a rate (and a table built
    // by mpcx_prg_ckpt_build has been given its length), and a stream that is read again from the start -- fresh engines
    // with the same keys -- has its table already.
    {
        const bool advancing = T.last_need != 0 && need > T.last_need && !save_now;
        const u64 step = advancing ? need - T.last_need : 0;
        if (need > T.last_need) T.last_need = need;
        if (advancing) {
            u64 more = 4 * step;
            const u64 lo = (16ull << 20) / PRG_CK, hi = (256ull << 20) / PRG_CK;
            more = more < lo ? lo : more > hi ? hi : more;
            if (T.host.size() < need + more / 2) walk_ahead(T, Key{ctx->device, seed, lcg}, need + more);
        }
    }
    if (T.dev_count < need) {
        MPCX_HIP(hipSetDevice(ctx->device));
        MPCX_HIP(upload(T, ctx->device));
    }
    *dev = T.dev;
    return 0;
}
int mpcx_prg_table(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_end, const PrgCk **dev) {
    return prg_table(ctx, seed, lcg, t_end, dev, false);
}

// the last context of a device has shut down: no kernel can read the superseded device copies any more
void prg_tables_context_opened(int device) {
    Registry &R = reg();
    std::lock_guard<std::mutex> g(R.mu);
    R.live_contexts[device]++;
}
void prg_tables_context_closed(int device) {
    Registry &R = reg();
    std::lock_guard<std::mutex> g(R.mu);
    if (--R.live_contexts[device] > 0) return;
    for (auto &kv : R.tables)
        if (kv.first.device == device) {
            CallerLock t(kv.second);
            for (PrgCk *p : kv.second.retired) (void)hipFree(p);
            kv.second.retired.clear();
        }
}

static int prg_launch(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_first, u64 t_end, int mode, u64 *out64, u8 *out8) {
    const PrgCk *ck;
    MPCX_TRY(mpcx_prg_table(ctx, seed, lcg, t_end, &ck));
    const u64 seg0 = t_first / PRG_CK, seg1 = (t_end - 1) / PRG_CK;
    const u64 nseg = seg1 - seg0 + 1;
    const u64 grid = ceil_div(nseg, 256);
    MPCX_REQUIRE(grid < (1ull << 31), "prg fill: range too large for one launch");
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(prg_fill_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, ck, seg0, nseg, t_first, t_end, mode,
                       out64, out8);
    MPCX_LAUNCH_CHECK();
    return 0;
}

// words at stream positions t_first + stride*i, i in [0,count) -> out[i]; with stride 2 the odd
// positions' low bits -> bools[i]
int prg_fill_words(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_first, u64 count, u64 stride, i64 *out, i8 *bools) {
    if (count == 0) return 0;
    MPCX_REQUIRE(stride == 1 || stride == 2, "prg_fill_words: stride must be 1 or 2");
    MPCX_REQUIRE(out || bools, "prg_fill_words: no output buffer");
    MPCX_REQUIRE(stride == 2 || out, "prg_fill_words: pop fill needs an output buffer");
    return prg_launch(ctx, seed, lcg, t_first, t_first + stride * count, stride == 1 ? 0 : 1, (u64 *)out, (u8 *)bools);
}

// out[i] = (X ? X[i] : 0) + streamA[tA_first + i] - streamB[tB_first + i], i in [0, count): the mask
// pop(next) - pop(prev) that the ABY3 share and product segments add (ABY3Engine.cpp:70-74, :190-197), without
// materialising either stream
int prg_fill_diff(mpcx_ctx *ctx, u64 seedA, u64 tA_first, u64 seedB, u64 tB_first, u64 count, const i64 *X, i64 *out) {
    if (count == 0) return 0;
    MPCX_REQUIRE(out, "prg_fill_diff: out is NULL");
    const u64 *x = (const u64 *)X;
    u64 *o = (u64 *)out;
    auto diff = [] __device__(u64 a, u64 b) { return a - b; };
    if (x) return prg_map2(ctx, seedA, tA_first, seedB, tB_first, count, diff, [=] __device__(u64 i, u64 v) { o[i] = x[i] + v; });
    return prg_map2(ctx, seedA, tA_first, seedB, tB_first, count, diff, [=] __device__(u64 i, u64 v) { o[i] = v; });
}

// out[i] = (uint16)int_buff[first + i] % 67 (zero_to_one: 0 -> 1), one byte per value
int prg_fill_buff_mod67(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 count, int zero_to_one, u8 *out) {
    if (count == 0) return 0;
    MPCX_REQUIRE(out, "prg_fill_buff_mod67: out is NULL");
    if ((2 * first) % PRG_CK == 0 && count % (PRG_CK / 2) == 0 && ((uintptr_t)out & 15) == 0) {
        // the run starts on a checkpoint and ends on one (every caller asks for 64 bytes per element): register-only path
        const PrgCk *ck;
        MPCX_TRY(mpcx_prg_table(ctx, seed, lcg, 2 * (first + count) + PRG_CK, &ck));
        const u64 grid = ceil_div(ceil_div(count, PRG_CK / 2), 256);
        MPCX_REQUIRE(grid < (1ull << 31), "prg fill: range too large for one launch");
        ProfScope ps(ctx, PROF_PRG);
        hipLaunchKernelGGL(prg_mod67_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, ck, 2 * first / PRG_CK, count,
                           zero_to_one, out);
        MPCX_LAUNCH_CHECK();
        return 0;
    }
    return prg_launch(ctx, seed, lcg, 2 * first, 2 * first + 2 * count - 1, zero_to_one ? 3 : 2, nullptr, out);
}

// out[64 e + k] = (uint16)int_buff[first + 64 e + k] % 67 (zero_to_one: 0 -> 1) for the *listed elements e of list
// only (device memory, filled by an earlier kernel on the stream); first is a multiple of 32, out 16-byte aligned
int prg_fill_buff_mod67_listed(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 n_elems, int zero_to_one, const u32 *list,
                               const u32 *listed, u8 *out) {
    if (n_elems == 0) return 0;
    MPCX_REQUIRE(out && list && listed, "prg_fill_buff_mod67_listed: NULL pointer");
    MPCX_REQUIRE((2 * first) % PRG_CK == 0 && ((uintptr_t)out & 15) == 0, "prg_fill_buff_mod67_listed: unaligned run");
    const PrgCk *ck;
    MPCX_TRY(mpcx_prg_table(ctx, seed, lcg, 2 * (first + 64 * n_elems) + PRG_CK, &ck));
    u64 grid = ceil_div(2 * n_elems, 256);
    if (grid > 2048) grid = 2048; // the lanes stride over the list, whose length only the device knows
    ProfScope ps(ctx, PROF_PRG);
    hipLaunchKernelGGL(prg_mod67_listed_kernel, dim3((u32)grid), dim3(256), 0, ctx->stream, ck, 2 * first / PRG_CK, list,
                       listed, zero_to_one, out);
    MPCX_LAUNCH_CHECK();
    return 0;
}

void prg_rand_cache_clear(mpcx_ctx *ctx) {
    if (ctx->rand_cache.empty()) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->rand_cache) (void)hipFree(e.dev);
    ctx->rand_cache.clear();
    ctx->rand_bytes = 0;
}

int prg_buff_mod67_cached(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 count, int zero_to_one, u64 ws_total,
                          u64 ws_offset, const u8 **out, int *left_to_caller) {
    if (left_to_caller) *left_to_caller = 0;
    const int mode = zero_to_one ? 3 : 2;
    const u64 epoch = ctx->rand_tick; // entries touched in this epoch belong to the call in progress
    for (auto &e : ctx->rand_cache)
        if (e.seed == seed && e.lcg == lcg && e.mode == mode && e.first <= first && first + count <= e.first + e.count &&
            (first - e.first) % 16 == 0) { // consumers read the bytes as aligned 8- and 16-byte words
            e.last_use = epoch;
            *out = e.dev + (first - e.first);
            return 0;
        }
    bool fits = count <= ctx->rand_cap;
    while (fits && ctx->rand_bytes + count > ctx->rand_cap) {
        int victim = -1;
        for (int i = 0; i < (int)ctx->rand_cache.size(); i++)
            if (ctx->rand_cache[i].last_use != epoch && (victim < 0 || ctx->rand_cache[i].last_use < ctx->rand_cache[victim].last_use))
                victim = i;
        if (victim < 0) {
            fits = false;
            break;
        }
        MPCX_HIP(hipStreamSynchronize(ctx->stream)); // a kernel in flight may still read the victim
        MPCX_HIP(hipFree(ctx->rand_cache[victim].dev));
        ctx->rand_bytes -= ctx->rand_cache[victim].count;
        ctx->rand_cache.erase(ctx->rand_cache.begin() + victim);
    }
    if (!fits) {
        MPCX_REQUIRE(ws_offset + count <= ws_total, "prg_buff_mod67_cached: scratch range outside the workspace");
        void *ws;
        MPCX_TRY(mpcx_ws(ctx, ws_total, &ws));
        u8 *fallback = (u8 *)ws + ws_offset;
        if (left_to_caller)
            *left_to_caller = 1;
        else
            MPCX_TRY(prg_fill_buff_mod67(ctx, seed, lcg, first, count, zero_to_one, fallback));
        *out = fallback;
        return 0;
    }
    u8 *dev = nullptr;
    MPCX_HIP(hipSetDevice(ctx->device));
    MPCX_HIP(hipMalloc((void **)&dev, count));
    int rc = prg_fill_buff_mod67(ctx, seed, lcg, first, count, zero_to_one, dev);
    if (rc) {
        (void)hipFree(dev);
        return rc;
    }
    ctx->rand_cache.push_back({seed, lcg, first, count, mode, dev, epoch});
    ctx->rand_bytes += count;
    *out = dev;
    return 0;
}

extern "C" {

int mpcx_set_rand_cache(mpcx_ctx *ctx, uint64_t max_bytes) {
    if (max_bytes < ctx->rand_bytes) prg_rand_cache_clear(ctx);
    ctx->rand_cap = max_bytes;
    return 0;
}

int mpcx_prg_ckpt_build(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t n_pops, uint64_t n_buff) {
    u64 t_end = 2 * n_buff;
    if (n_pops && PRG_POP_T0 + n_pops > t_end) t_end = PRG_POP_T0 + n_pops;
    const PrgCk *ck;
    return prg_table(ctx, seed, lcg, t_end, &ck, true);
}

int mpcx_prg_fill_buff(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t first, uint64_t count, int64_t *ints,
                       int8_t *bools) {
    return prg_fill_words(ctx, seed, lcg, 2 * first, count, 2, ints, bools);
}

int mpcx_prg_fill_pop(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t first_pop, uint64_t count, int64_t *out) {
    MPCX_REQUIRE(out || count == 0, "mpcx_prg_fill_pop: out is NULL");
    return prg_fill_words(ctx, seed, lcg, PRG_POP_T0 + first_pop, count, 1, out, nullptr);
}

int mpcx_prg_state_at_pop(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t pop_index, uint64_t state_out[3]) {
    const u64 t = PRG_POP_T0 + pop_index;
    const PrgCk *ck;
    MPCX_TRY(mpcx_prg_table(ctx, seed, lcg, t, &ck));
    PrgCk c;
    {
        Table &T = table_of(ctx->device, seed, lcg);
        CallerLock g(T);
        c = T.host[t / PRG_CK];
    }
    for (u64 i = (t / PRG_CK) * PRG_CK; i < t; i++) prv_step(c.seed, c.lcg, c.hash);
    state_out[0] = c.seed;
    state_out[1] = c.lcg;
    state_out[2] = c.hash;
    return 0;
}

} // extern "C"
