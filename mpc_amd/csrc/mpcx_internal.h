// Internal declarations shared by the HIP translation units of libmpcx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpcx.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint16_t u16;
typedef int8_t i8;
typedef uint8_t u8;

// One checkpoint of a prvhash stream: the generator state before stream word t = PRG_CK * index,
// where stream word t is raw output #(5 + t) of PRgenerator(seed, lcg) (utils/PRgenerator.cpp:21-33).
struct PrgCk {
    u64 seed, lcg, hash;
};
// Stream words per checkpoint = words one lane regenerates.  64 keeps a 2^24-word fill at four waves per SIMD
// (256 left it at one, and the dependent multiply chain then ran at 40 % of its issue rate); the table costs
// 0.375 bytes per stream word.
constexpr u64 PRG_CK = 64;
constexpr u64 PRG_POP_T0 = 320000; // stream word of the first pop (2 * BUFF_NUM, utils/PRgenerator.hpp:6)

// optional per-kernel timing with HIP events on the context's stream (mpcx_profile / mpcx_profile_read)
enum ProfSlotId { PROF_GEMM = 0, PROF_SPLIT, PROF_PRG, PROF_PC_PARTY, PROF_PC_HELPER, PROF_MSB, PROF_SLOTS };

struct mpcx_ctx {
    bool prof = false;
    unsigned prof_mask = ~0u; // slots that record (mpcx_profile_select)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[PROF_SLOTS];
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    void *stage[2] = {nullptr, nullptr}; // pinned staging buffers of the large host <-> device copies
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    void *ws = nullptr; // scratch workspace, grown on demand
    u64 ws_bytes = 0;
    // Values derived from a generator's int_buff, kept on the device between calls.  The reference fills
    // int_buff once in the PRgenerator constructor (utils/PRgenerator.cpp:27-32) and every later call only
    // reads it; this cache is the counterpart for the slices a kernel consumes as bytes mod 67.
    struct RandEntry {
        u64 seed, lcg, first, count;
        int mode;
        u8 *dev;
        u64 last_use;
    };
    std::vector<RandEntry> rand_cache;
    u64 rand_bytes = 0, rand_cap = 4ull << 30, rand_tick = 0;
    // mpcx_malloc / mpcx_free recycle device blocks: every use of a block is ordered on the context's stream,
    // so a freed block can be handed out again without waiting for the kernels that still read it.
    // mpcx_aby3_msb_begin: the G0, G1, P0, P1 arrays that the first round of a carry tree reads in place, per state buffer
    std::map<const void *, std::array<const i64 *, 4>> msb_inputs;
    std::multimap<u64, void *> pool_free; // idle blocks by size
    std::map<void *, u64> pool_live;      // blocks handed out -> size
    u64 pool_idle_bytes = 0, pool_cap = 8ull << 30;
};

int mpcx_fail(mpcx_ctx *ctx, const char *fmt, ...);
// memcpy cut into pieces moved by the library's pool of host threads (MPCX_COPY_THREADS; default half the hardware threads, 4 to 16) and the caller
void mpcx_parallel_memcpy(void *dst, const void *src, u64 bytes);
// strided runs of pageable host memory -> contiguous device memory, through the pinned staging buffers (ctx.hip)
int mpcx_h2d_gather(mpcx_ctx *ctx, void *dst, const void *src, u64 spitch, u64 width, u64 height);
// scratch of at least `bytes`, 256-byte aligned, valid until the next mpcx_ws call on this context
int mpcx_ws(mpcx_ctx *ctx, u64 bytes, void **out);
// device pointer to the checkpoint table of (seed,lcg) covering stream words [0, t_end)
int mpcx_prg_table(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_end, const PrgCk **dev);

#define MPCX_HIP(call)                                                                                                 \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return mpcx_fail(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define MPCX_TRY(call)                                                                                                 \
    do {                                                                                                               \
        int r_ = (call);                                                                                               \
        if (r_) return r_;                                                                                             \
    } while (0)
#define MPCX_REQUIRE(cond, ...)                                                                                        \
    do {                                                                                                               \
        if (!(cond)) return mpcx_fail(ctx, __VA_ARGS__);                                                               \
    } while (0)
#define MPCX_LAUNCH_CHECK() MPCX_HIP(hipGetLastError())

// records a start/stop event pair around the launches issued during its lifetime (no-op unless profiling)
struct ProfScope {
    mpcx_ctx *ctx;
    int slot;
    hipEvent_t stop = nullptr;
    ProfScope(mpcx_ctx *c, int s) : ctx(c), slot(s) {
        if (!ctx->prof || !((ctx->prof_mask >> s) & 1u)) return;
        hipEvent_t start;
        if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
            stop = nullptr;
            return;
        }
        (void)hipEventRecord(start, ctx->stream);
        ctx->prof_ev[slot].push_back({start, stop});
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, ctx->stream);
    }
};

// internal launchers used across translation units (all asynchronous on ctx->stream)
int prg_fill_words(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 t_first, u64 count, u64 stride, i64 *out, i8 *bools);
int prg_fill_buff_mod67(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 count, int zero_to_one, u8 *out);
// out[i] = (X ? X[i] : 0) + A[tA_first + i] - B[tB_first + i] over the stream words of generators (seedA, 0), (seedB, 0)
int prg_fill_diff(mpcx_ctx *ctx, u64 seedA, u64 tA_first, u64 seedB, u64 tB_first, u64 count, const i64 *X, i64 *out);
// The same bytes through the context's cache: *out points at cached device memory.  Entries requested since
// the caller last advanced ctx->rand_tick are never evicted, so a call bumps the tick once and may then hold
// several slices.  A slice that does not fit the cache budget is generated at byte ws_offset of the context
// workspace, which is grown to ws_total bytes (pass the same ws_total for every request of one call) -- or, with
// left_to_caller non-null, is NOT generated: *left_to_caller becomes 1 and the caller fills the part it needs.
int prg_buff_mod67_cached(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 count, int zero_to_one, u64 ws_total,
                          u64 ws_offset, const u8 **out, int *left_to_caller = nullptr);
// out[64 e + k] = (uint16)int_buff[first + 64 e + k] % 67 for the *listed elements e of list only (both in device
// memory, written by an earlier kernel on the stream)
int prg_fill_buff_mod67_listed(mpcx_ctx *ctx, u64 seed, u64 lcg, u64 first, u64 n_elems, int zero_to_one, const u32 *list,
                               const u32 *listed, u8 *out);
void prg_rand_cache_clear(mpcx_ctx *ctx);
// the checkpoint tables of a device keep their superseded device copies while any context of the device is open
void prg_tables_context_opened(int device);
void prg_tables_context_closed(int device);
// Z_67 sharing of the 64 bits of v[i]: first[i][k] = rnd[64 i + k] (one byte, already % 67),
// second[i][k] = (bit_k(v[i]) + 67 - first[i][k]) % 67; first / second are 16-byte aligned
int snn_bit_share_launch(mpcx_ctx *ctx, const u64 *v, const u8 *rnd, u64 n, u16 *first, u16 *second);
static inline u64 ceil_div(u64 a, u64 b) { return (a + b - 1) / b; }
#if defined(__HIPCC__)
// 16-byte accesses with the streaming (non-temporal) hint, for messages and share arrays that a kernel touches once:
// on the gigabyte-sized privateCompare / bit-sharing vectors they measured 2-5 % faster than plain accesses.  Not for
// data the next kernel reads back while it may still sit in the last-level cache: on the messages of the ABY3 msb
// rounds (halving in size every round) the hint cost 9 %.
typedef u32 mpcx_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint4 ld_stream(const uint4 *p) {
    const mpcx_u32x4 v = __builtin_nontemporal_load((const mpcx_u32x4 *)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline void st_stream(uint4 *p, u32 a, u32 b, u32 c, u32 d) {
    const mpcx_u32x4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, (mpcx_u32x4 *)p);
}
#endif
// Largest grid the streaming kernels are launched with.  They all grid-stride, but for the privateCompare and
// element-wise kernels a launch with one item per lane (hundreds of thousands of small workgroups) streams
// measurably faster on MI355X than a capped, looping one (privateCompare helper at 2^24 elements: 6.2 vs
// 4.9 TB/s), so their cap is only the API limit; kernels that measured otherwise keep their own smaller cap.
// The environment variable MPCX_MAX_GRID lowers the limit, which is how the tests exercise the loops.
static inline u64 mpcx_max_grid() {
    static const u64 v = [] {
        const char *e = getenv("MPCX_MAX_GRID");
        const u64 x = e ? strtoull(e, nullptr, 10) : 0;
        return x ? x : 0x7FFFFFFFull;
    }();
    return v;
}
#define MPCX_MAX_GRID mpcx_max_grid()

// Z (+)= [A_1 | A_2 | ..] * [B_1 (+Bplus_1) ; B_2 ; ..] (+ add - sub): a ring product whose K dimension is the
// concatenation of `nparts` (A_p : r x l, B_p : l x c) pairs.  add/sub are row-major r*c word arrays.
struct GemmPart {
    const i64 *A;
    const i64 *B, *Bplus; // Bplus (nullable) is added element-wise to B before the limb split
};
struct RnsPreB;
int ring_gemm_parts(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                    const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, const RnsPreB *pre = nullptr);
// workspace bytes ring_gemm_parts takes from the front of the context workspace (a caller may reserve
// more and use the tail for itself)
u64 ring_gemm_ws_bytes(int nparts, u64 r, u64 l, u64 c);
// the same product through a residue number system (ring_gemm_rns.hip): 18 plain int8 products instead of 36
bool ring_gemm_use_rns(int nparts, u64 r, u64 l, u64 c);
bool ring_gemm_rns_wide(); // the residue form's block tile: 384 x 256 (default) or 256 x 256 (MPCX_RNS_TILE=0)
u32 gemm_group_rows(const char *env_name, u32 tiles_m);
// looks up, once per device, the address of the file-scope scratch array of the entry point that allocates nothing
int ring_gemm_static_scratch_init(int device);
u64 ring_gemm_rns_ws_bytes(int nparts, u64 r, u64 l, u64 c);
// The residue planes of a product's second operand, split once and kept: a product of any row count that runs in the
// residue form reads them instead of splitting B again (weights that many products use).  They belong to one
// (nparts, l, c, layout-independent) geometry and block tile; a product they do not fit splits B as usual.
struct RnsPreB {
    i8 *planes = nullptr; // nchunks x chunk_bytes, one K chunk after the other
    u64 chunk_bytes = 0, nchunks = 0;
    int nparts = 0;
    u64 l = 0, c = 0;
    bool wide = false;
    mutable u64 used = 0; // products that read the planes
};
int ring_gemm_rns_prepare_b(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 l, u64 c, int layB, RnsPreB *out);
void ring_gemm_rns_release_b(mpcx_ctx *ctx, RnsPreB *pre);
int ring_gemm_parts_rns(mpcx_ctx *ctx, const GemmPart *parts, int nparts, u64 r, u64 l, u64 c, int layA, int layB,
                        const i64 *add, const i64 *sub, i64 *Z, int layZ, int accumulate, const RnsPreB *pre = nullptr);
