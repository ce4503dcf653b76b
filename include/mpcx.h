/* mpcx -- C ABI of the MI355X (gfx950) implementation of the per-party local arithmetic of
 * victerying/mpc's ABY3Engine, SecurennEngine and MyEngine.
 *
 * The reference has no FFI: its boundary is the C++ engine classes
 * (mpc_engine/ABY3Engine.hpp:24-54, mpc_engine/SecurennEngine.hpp:20-50) whose methods all have the
 * shape  local arithmetic -> VirtuleMessenger::send_and_recv -> local arithmetic.  This header exports
 * ONE entry point per local-arithmetic segment between two messenger rounds.  Every message buffer
 * uses exactly the reference's wire layout (utils/util.hpp:19-105,135-154,184-200: little-endian
 * 64-bit words, matrices row-major, Z_67 bit shares as 64 uint16 per element LSB first, booleans one
 * byte each), so a host engine built on these calls interoperates with reference peers.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer (hipMalloc / mpcx_malloc / a torch tensor's data_ptr),
 *    8-byte aligned unless it holds bytes; sizes and indices are uint64_t element counts;
 *  - calls are asynchronous on the context's HIP stream (mpcx_set_stream / mpcx_sync);
 *  - return value: 0 = ok, non-zero = error (text from mpcx_last_error); shapes are validated on the
 *    host before any kernel is launched;
 *  - one context per engine / host thread; a context is not thread-safe; its calls are made with its device
 *    current on the calling thread (mpcx_init leaves it so; a thread that drives contexts of several devices calls
 *    hipSetDevice before switching, or uses the mpcx_group_* calls, whose worker threads are bound to their devices);
 *  - PRG streams are named by the PRgenerator seed (lcg is 0 in both engines,
 *    mpc_engine/ABY3Engine.cpp:34-53).  "pop index" k = the k-th pop_int64()/pop_bool() after
 *    construction (utils/PRgenerator.cpp:34-42); "buffer index" i = int_buff[i] / bool_buff[i]
 *    (utils/PRgenerator.cpp:27-32), defined for every i >= 0 by the same recurrence (the reference only
 *    stores i < BUFF_NUM = 160000, utils/PRgenerator.hpp:6);
 *  - (n_total, i0, n): the call handles elements [i0, i0+n) of a logical vector of n_total elements
 *    (buffer offsets in SecurennEngine depend on the whole vector's size), which is how element
 *    ranges are sharded over GPUs;
 *  - layout flags: 0 = row-major, 1 = column-major (how Eigen stores MatrixXl).
 */
#ifndef MPCX_H
#define MPCX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcx_ctx mpcx_ctx;

/* ---- context, memory, errors (no reference counterpart: the reference is host-only) ---- */
int mpcx_init(int device, mpcx_ctx **out);
int mpcx_shutdown(mpcx_ctx *ctx);
const char *mpcx_last_error(mpcx_ctx *ctx); /* ctx may be NULL: error of a failed mpcx_init */
int mpcx_set_stream(mpcx_ctx *ctx, void *hip_stream);
void *mpcx_get_stream(mpcx_ctx *ctx);
int mpcx_sync(mpcx_ctx *ctx);
/* device memory.  Freed blocks are recycled by later mpcx_malloc calls of the same context (all work of a
 * context is ordered on its stream, so mpcx_free never waits); at most max_idle_bytes (default 8 GiB) are
 * kept idle, the rest goes back to the driver. */
int mpcx_malloc(mpcx_ctx *ctx, uint64_t bytes, void **out);
int mpcx_free(mpcx_ctx *ctx, void *p);
int mpcx_set_pool_limit(mpcx_ctx *ctx, uint64_t max_idle_bytes);
int mpcx_h2d(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int mpcx_d2h(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* Boolean payloads: nbits booleans that the device holds as a bit string (bit b of byte b / 8) <-> nbits bytes of 0 / 1
 * in host memory (the layout of vecbool2string, utils/util.hpp).  d2h_bits reads ceil(nbits / 8) device bytes;
 * h2d_bits takes bit 0 of every host byte and writes whole 32-bit words, the bits beyond nbits zero. */
int mpcx_d2h_bits(mpcx_ctx *ctx, void *dst_bytes, const void *src_bits, uint64_t nbits);
int mpcx_h2d_bits(mpcx_ctx *ctx, void *dst_bits, const void *src_bytes, uint64_t nbits);
/* `count` uint16 values between a device array and (possibly unaligned) host memory, e.g. the Z_67 shares of
 * privateCompare (SecurennEngine.cpp:559-589: uint16 on the wire, values below 67).  Same result as mpcx_d2h / mpcx_h2d
 * of 2 * count bytes; when no value exceeds 255 only count bytes cross the host link (narrowed / widened by a kernel on
 * the device and by the copy threads on the host), otherwise the array goes as it is. */
int mpcx_d2h_u16(mpcx_ctx *ctx, void *dst, const void *src, uint64_t count);
int mpcx_h2d_u16(mpcx_ctx *ctx, void *dst, const void *src, uint64_t count);
int mpcx_d2d(mpcx_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* mpcx_h2d / mpcx_d2h return when the copy is complete; host memory may be pageable (the std::string payloads of
 * VirtuleMessenger::send_and_recv, utils/VirtuleMessenger.hpp:14).  Transfers of 4 MiB and more are pipelined through
 * two pinned staging buffers of the context, DMA on one chunk while host threads (MPCX_COPY_THREADS; default: three quarters of the hardware threads, 4 to 16) copy
 * the other.  mpcx_host_copy is that multi-threaded memcpy by itself, for messengers that move payloads between
 * host buffers (utils/SocketMessenger.cpp:142-164 does it through the kernel's socket buffers). */
int mpcx_host_copy(void *dst, const void *src, uint64_t bytes);
/* dst (cols x rows, row-major) = the transpose of src (rows x cols, row-major), 8-byte words, host memory, on the same
 * threads: the conversion between the row-major wire order of matrix2string / string2matrix (utils/util.hpp:19-52) and
 * Eigen's column-major storage.  dst and src must not overlap.  Needs no context and no GPU. */
int mpcx_host_transpose64(void *dst, const void *src, uint64_t rows, uint64_t cols);
/* per-kernel device timing with HIP events recorded on the context's stream: mpcx_profile(ctx,1) starts
 * (and clears) the collection, mpcx_profile_read sums the elapsed time of every launch of one kernel
 * family since then: "limb_gemm", "split_limbs", "prg_fill", "pc_party", "pc_helper",
 * "msb_round" */
int mpcx_profile(mpcx_ctx *ctx, int enable);
/* restricts the collection to one kernel family (NULL: all again).  Every recorded event costs the stream a few
 * microseconds between two kernels, so a timed region that needs one kernel's durations asks for that one only. */
int mpcx_profile_select(mpcx_ctx *ctx, const char *kernel);
int mpcx_profile_read(mpcx_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);
/* grows the context's scratch workspace up front so that later calls do not allocate */
int mpcx_reserve_workspace(mpcx_ctx *ctx, uint64_t bytes);

/* ---- PRgenerator (utils/PRgenerator.cpp:3-42) ---- */
/* Walks the stream on the host once and keeps a checkpoint every 64 outputs (24 bytes each) on host and
 * device, so that kernels can regenerate any range in parallel; the table is shared by the contexts of a device.  n_pops / n_buff: how far the pop stream / the
 * buffers will be read.  Called implicitly (and lazily) by everything below. */
int mpcx_prg_ckpt_build(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t n_pops, uint64_t n_buff);
/* int_buff[first .. first+count) -> ints (nullable), bool_buff[...] -> bools (nullable)   (:27-32) */
int mpcx_prg_fill_buff(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t first, uint64_t count, int64_t *ints,
                       int8_t *bools);
/* pops first_pop .. first_pop+count of pop_int64()                                         (:34-37) */
int mpcx_prg_fill_pop(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t first_pop, uint64_t count, int64_t *out);
/* The reference fills int_buff once, in the PRgenerator constructor (:27-32), and privateCompare,
 * shareConvert and msb_L_1 afterwards only read it (SecurennEngine.cpp:523-534, :838-839, :922-923).  The
 * library's counterpart: the "(uint16)int_buff[i] % 67" byte slices those calls consume are generated on
 * first use and kept on the device, up to max_bytes per context (default 4 GiB; 64 bytes per element and
 * slice; least recently used slices are dropped; 0 = regenerate on every call). */
int mpcx_set_rand_cache(mpcx_ctx *ctx, uint64_t max_bytes);
/* host-side: the (seed, lcg, hash) members of a generator that has popped pop_index words  (:12-14) */
int mpcx_prg_state_at_pop(mpcx_ctx *ctx, uint64_t seed, uint64_t lcg, uint64_t pop_index, uint64_t state_out[3]);

/* ---- Z_2^64 helpers (Eigen int64 operator+,-,* as the engines use them) ---- */
/* C (+)= A*B mod 2^64 on the int8 matrix cores (exact: signed base-256 digits, or residues + CRT); A is r x l, B is l x c */
int mpcx_ring_gemm_u64(mpcx_ctx *ctx, const int64_t *A, const int64_t *B, int64_t *C, uint64_t r, uint64_t l,
                       uint64_t c, int layoutA, int layoutB, int layoutC, int accumulate);
/* int8 products the library issues per ring multiply-add of an r x l . l x c product: 36 (digit pairs of the
 * positional form) or 18 (residues modulo 18 coprime moduli + CRT), chosen by an estimate of both run times; MPCX_GEMM_RNS=0 / 2 in the
 * environment forces the first / the second.  No counterpart in the reference (Eigen's operator*). */
int mpcx_ring_gemm_products(uint64_t r, uint64_t l, uint64_t c);
/* Name of the GEMM kernel such a product runs in ("limb_gemm_kernel", "rns_gemm_kernel": 256 x 256 block tiles,
 * "rns_gemm_wide_kernel": 384 x 256, the default of the residue form; MPCX_RNS_TILE=0 selects the former) -- for
 * profilers that match kernel names.  The string is static.  No counterpart in the reference. */
const char *mpcx_ring_gemm_kernel_name(uint64_t r, uint64_t l, uint64_t c);
/* mpcx_ring_gemm_u64 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results; the limb
 * planes live in a file-scope device array of mpcx_ring_gemm_noalloc_limit() = 16 MiB per GPU (held until the process
 * ends) instead of the context workspace, and the call makes no hipMalloc / hipFree and creates no events.  Always
 * the digit-pair kernels.  A shape needs mpcx_ring_gemm_noalloc_bytes(r, l, c) bytes of planes: 8 * (ceil128(r) +
 * ceil128(c)) * kc, where kc = ceil64(l) halved (and rounded up to 64 again) until it is at most 8192, the length of
 * one accumulation chunk; shapes that need more than the limit are rejected with an error, e.g. 1024 x 1024 x 1024 and
 * 128 x 8192 x 128 fit.  Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two
 * contexts, two streams) race. */
int mpcx_ring_gemm_u64_noalloc(mpcx_ctx *ctx, const int64_t *A, const int64_t *B, int64_t *C, uint64_t r, uint64_t l,
                               uint64_t c, int layoutA, int layoutB, int layoutC, int accumulate);
uint64_t mpcx_ring_gemm_noalloc_limit(void);
uint64_t mpcx_ring_gemm_noalloc_bytes(uint64_t r, uint64_t l, uint64_t c);
int mpcx_ring_add(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n);
int mpcx_ring_sub(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n);
int mpcx_ring_sar(mpcx_ctx *ctx, const int64_t *a, int64_t *out, uint64_t n, int k); /* benchmark/benchmark1.cpp:106-112 */
int mpcx_ring_shl1(mpcx_ctx *ctx, const int64_t *a, int64_t *out, uint64_t n);       /* SecurennEngine.cpp:1094-1098 */
int mpcx_relayout(mpcx_ctx *ctx, const int64_t *src, int64_t *dst, uint64_t rows, uint64_t cols, int lay_src,
                  int lay_dst);

/* ---- ABY3Engine (mpc_engine/ABY3Engine.cpp) ---- */
/* share / share_matrix before the round (:63-78, :83-103): out[i] = pop(next) - pop(prev) (+ X[i] when
 * X != NULL, i.e. on party 0), i in row-major order */
int mpcx_aby3_share_local(mpcx_ctx *ctx, const int64_t *X, uint64_t n, uint64_t seed_next, uint64_t pop_next,
                          uint64_t seed_prev, uint64_t pop_prev, int64_t *out);
/* share_bools before the round (:108-124) */
int mpcx_aby3_share_bools_local(mpcx_ctx *ctx, const int8_t *X, uint64_t n, uint64_t seed_next, uint64_t pop_next,
                                uint64_t seed_prev, uint64_t pop_prev, int8_t *out);
/* matrix_mul before the round (:189-199): Z0 = X0*Y0 + X0*Y1 + X1*Y0 + (pop(next) - pop(prev)), inputs in
 * `layout`, Z0 row-major (= matrix2string's order) */
int mpcx_aby3_matmul_local(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                           uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next, uint64_t pop_next,
                           uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0);
/* row-block form for sharding over GPUs: rows [row0, row0+rows) of the r x l operands; X0/X1 point at the
 * block (rows x l, row-major), Z0 receives rows x c; the mask uses pops pop_* + row0*c ... */
int mpcx_aby3_matmul_local_rows(mpcx_ctx *ctx, const int64_t *X0blk, const int64_t *X1blk, const int64_t *Y0,
                                const int64_t *Y1, uint64_t rows, uint64_t l, uint64_t c, uint64_t row0,
                                uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                                int64_t *Z0blk);
/* The same product with a second operand that many calls use (weights): matrix_mul (:189-197) reads Y0 + Y1 and Y0
 * afresh in every call; here the pair stays on the device where the caller keeps it (it must not change or go away
 * before mpcx_aby3_operand_free) and, for shapes whose products run as 18 residue products, the residue planes of
 * [Y0 + Y1 ; Y0] are made once by mpcx_aby3_operand_prepare (mpcx_aby3_operand_bytes tells how much device memory
 * they take; 0 = none were made) and read by every mpcx_aby3_matmul_local_prepared.  X0 / X1: rows x l in layoutX
 * (rows [row0, row0 + rows) of the whole operand, as in the row-block form), Z0: rows x c row-major.  Results are those
 * of mpcx_aby3_matmul_local_rows, word for word. */
typedef struct mpcx_aby3_operand mpcx_aby3_operand;
int mpcx_aby3_operand_prepare(mpcx_ctx *ctx, const int64_t *Y0, const int64_t *Y1, uint64_t l, uint64_t c, int layout,
                              mpcx_aby3_operand **out);
int mpcx_aby3_operand_free(mpcx_ctx *ctx, mpcx_aby3_operand *op);
uint64_t mpcx_aby3_operand_bytes(const mpcx_aby3_operand *op);
uint64_t mpcx_aby3_operand_uses(const mpcx_aby3_operand *op); /* products so far that read the planes instead of splitting */
int mpcx_aby3_matmul_local_prepared(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, int layoutX,
                                    const mpcx_aby3_operand *Y, uint64_t rows, uint64_t row0, uint64_t seed_next,
                                    uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0);
/* vector_mul before the round (:220-227) */
int mpcx_aby3_vecmul_local(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *Y0, const int64_t *Y1,
                           uint64_t n, uint64_t seed_next, uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev,
                           int64_t *Z0);
/* vecbool_mul before the round (:253-256): Z0 = (X0&Y0)^(X1&Y0)^(X0&Y1), one byte per boolean */
int mpcx_aby3_vecbool_mul_local(mpcx_ctx *ctx, const int8_t *X0, const int8_t *X1, const int8_t *Y0, const int8_t *Y1,
                                uint64_t n, int8_t *Z0);
/* reveal / reveal_matrix after the round (:142, :157): out = a + b + c ; reveal_bools (:171-174) and
 * reveal_bits (:287-288): out = a ^ b ^ c */
int mpcx_aby3_reveal3(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, const int64_t *c, int64_t *out, uint64_t n);
int mpcx_aby3_reveal3_bools(mpcx_ctx *ctx, const int8_t *a, const int8_t *b, const int8_t *c, int8_t *out, uint64_t n);
int mpcx_aby3_reveal3_bits(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, const int64_t *c, int64_t *out,
                           uint64_t n);
/* add_prepare (:295-326): step1 before round 1 (C0 = X0 & X1); step2 between the rounds */
int mpcx_aby3_add_prepare_step1(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, uint64_t n, int64_t *C0);
int mpcx_aby3_add_prepare_step2(mpcx_ctx *ctx, const int64_t *X0, const int64_t *X1, const int64_t *C0,
                                const int64_t *C1, uint64_t n, int64_t *P0, int64_t *P1, int64_t *G0);
/* msb (:361-467), six rounds temp = 32,16,8,4,2,1.  The carry tree is kept as dense bit strings (bit
 * 2*temp*k + b of a level's string = the reference's byte k*2*temp+b) that halve every round; `state` is an
 * opaque scratch of 6*n 64-bit words owned by the caller for the duration of one msb.
 *  begin:      records (in the context) where the inputs live; G0, G1, P0, P1 are read in place by the temp = 32
 *              round (with :385-387, G[63]=0, P[63]=1, applied on the fly) and must stay unchanged until its
 *              round_post / round_mid has run; the rounds of a state buffer go through the context that began it
 *  round_pre:  builds this round's vecbool_mul product share (:397-414, :253-256), writes it to msg_out
 *              (2*temp bytes per element, the byte order of result[0]) and advances share 0 of the state
 *  round_post: msg_in = result[1] received from next; advances share 1 of the state (:416-428)
 *  final:      Z_i = G_i ^ backup_i (:459-466); P0_in/P1_in are the msb() inputs (bit 63 = backup)
 * One call per local segment between two messenger rounds: round_pre(32) before the first round, then
 *  round_mid:  for temp = 32,16,8,4,2: msg_in = result[1] of round temp; the same state and msg_out as
 *              round_post(temp) followed by round_pre(temp / 2), in one pass (share 1 of the level in between
 *              stays in registers)
 *  last:       msg_in = result[1] of round 1; the same Z as round_post(1) followed by final (the backup bits are
 *              read from the state, where round_pre(32) keeps them) */
int mpcx_aby3_msb_begin(mpcx_ctx *ctx, const int64_t *G0, const int64_t *G1, const int64_t *P0, const int64_t *P1,
                        uint64_t n, uint64_t *state);
int mpcx_aby3_msb_round_pre(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, int8_t *msg_out);
int mpcx_aby3_msb_round_post(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in);
int mpcx_aby3_msb_final(mpcx_ctx *ctx, const uint64_t *state, const int64_t *P0_in, const int64_t *P1_in, uint64_t n,
                        int8_t *Z0, int8_t *Z1);
int mpcx_aby3_msb_round_mid(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const int8_t *msg_in,
                            int8_t *msg_out);
int mpcx_aby3_msb_last(mpcx_ctx *ctx, const uint64_t *state, uint64_t n, const int8_t *msg_in, int8_t *Z0, int8_t *Z1);
/* The same three segments with the messages kept as bit strings on the device: bit b of a string is wire byte b of the
 * vecbool2string payload (utils/util.hpp), ceil(bits / 32) words per message, bits beyond the message zero.  The wire
 * bytes are made from / packed into such strings on the host by mpcx_d2h_bits / mpcx_h2d_bits below, so an eighth of
 * the bytes cross the host link; state, results and the bytes on the wire are those of the byte forms. */
int mpcx_aby3_msb_round_pre_bits(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, uint32_t *bits_out);
int mpcx_aby3_msb_round_mid_bits(mpcx_ctx *ctx, uint64_t *state, uint64_t n, uint64_t temp, const uint32_t *bits_in,
                                 uint32_t *bits_out);
int mpcx_aby3_msb_last_bits(mpcx_ctx *ctx, const uint64_t *state, uint64_t n, const uint32_t *bits_in, int8_t *Z0,
                            int8_t *Z1);
/* trunc before the round (:480-503); seed12/pop12 = the stream parties 1 and 2 share */
int mpcx_aby3_trunc_local(mpcx_ctx *ctx, int party, const int64_t *X0, const int64_t *X1, uint64_t n, int number,
                          uint64_t seed12, uint64_t pop12, int64_t *Y0);
/* bit_inject (:510-1038), one pre (before the round) and one post (after) per party.  Message layouts:
 *  P0: to_p1 = (M0_0^W02_0)[n]|(M0_1^W02_1)[n]            to_p2 = (M0_0^W01_0)[n]|(M0_1^W01_1)[n]|W10_b[n]
 *  P1: to_p0 = (M1_0^W12_0)[n]|(M1_1^W12_1)[n]            to_p2 = W01_b[n]|(M1_0^W10_0)[n]|(M1_1^W10_1)[n]
 *  P2: to_p0 = W12_b[n]                                   to_p1 = W02_b[n]
 * pop01/pop20/pop12 = the caller's pop index into that stream when bit_inject starts (the call consumes
 * 6n words of s01 and 3n of s20 on P0; 6n of s01, 3n of s12 on P1; 3n of s20, 3n of s12 on P2).
 * A call handles elements [i0, i0+n) of a vector of n_total elements (one GPU's range; n_total = n, i0 = 0 for
 * the whole vector): the stream sections are indexed per whole vector (:539-558), the message sections above
 * have n entries each, so the ranges of a section concatenate to the whole vector's section. */
int mpcx_aby3_bit_inject_p0_pre(mpcx_ctx *ctx, const int64_t *A0, const int8_t *B0, const int8_t *B1, uint64_t n_total,
                                uint64_t i0, uint64_t n, uint64_t seed01, uint64_t pop01, uint64_t seed20,
                                uint64_t pop20, int64_t *to_p1, int64_t *to_p2);
int mpcx_aby3_bit_inject_p0_post(mpcx_ctx *ctx, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20,
                                 const int64_t *from_p1, const int64_t *from_p2, int64_t *C0, int64_t *C1);
int mpcx_aby3_bit_inject_p1_pre(mpcx_ctx *ctx, const int64_t *A0, const int64_t *A1, const int8_t *B0,
                                const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                                uint64_t pop01, uint64_t seed12, uint64_t pop12, int64_t *to_p0, int64_t *to_p2);
int mpcx_aby3_bit_inject_p1_post(mpcx_ctx *ctx, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                                 uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p2, int64_t *C0, int64_t *C1);
int mpcx_aby3_bit_inject_p2_pre(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                int64_t *to_p0, int64_t *to_p1);
int mpcx_aby3_bit_inject_p2_post(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                 uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                 const int64_t *from_p0, const int64_t *from_p1, int64_t *C0, int64_t *C1);
/* The ReLU composition x - bit_inject(x, msb(add_prepare(x))) (benchmark/benchmark1.cpp:71-73 followed by the
 * subtraction of the two share vectors): the post segment of bit_inject fused with that subtraction.  Same
 * arguments as the post call of the party plus the minuend shares S0 / S1; R_k = S_k - C_k, where C_k is what the
 * post call would have written.  R may be the same array as S. */
int mpcx_aby3_bit_inject_p0_post_sub(mpcx_ctx *ctx, const int8_t *B0, uint64_t n_total, uint64_t i0, uint64_t n,
                                     uint64_t seed01, uint64_t pop01, uint64_t seed20, uint64_t pop20,
                                     const int64_t *from_p1, const int64_t *from_p2, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1);
int mpcx_aby3_bit_inject_p1_post_sub(mpcx_ctx *ctx, const int8_t *B1, uint64_t n_total, uint64_t i0, uint64_t n,
                                     uint64_t seed01, uint64_t pop01, uint64_t seed12, uint64_t pop12,
                                     const int64_t *from_p0, const int64_t *from_p2, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1);
int mpcx_aby3_bit_inject_p2_post_sub(mpcx_ctx *ctx, const int8_t *B0, const int8_t *B1, uint64_t n_total, uint64_t i0,
                                     uint64_t n, uint64_t seed20, uint64_t pop20, uint64_t seed12, uint64_t pop12,
                                     const int64_t *from_p0, const int64_t *from_p1, const int64_t *S0,
                                     const int64_t *S1, int64_t *R0, int64_t *R1);

/* ---- SecurennEngine (mpc_engine/SecurennEngine.cpp) ---- */
/* share (:318-334): party 0: -pop(s01); party 1: X + pop(s01); party 2: zeros */
int mpcx_snn_share_local(mpcx_ctx *ctx, int party, const int64_t *X, uint64_t n, uint64_t seed01, uint64_t pop01,
                         int64_t *out);
/* reveal_L_1 (:448,:459) and reveal_zp (:482,:496) combine steps */
int mpcx_snn_add_L1(mpcx_ctx *ctx, const int64_t *a, const int64_t *b, int64_t *out, uint64_t n);
int mpcx_snn_add_zp(mpcx_ctx *ctx, const uint16_t *a, const uint16_t *b, uint16_t *out, uint64_t n64);
/* matrix_mul (:61-169).  dealer = party 2 before round 1: msg_i = A_i|B_i|C_i (row-major) from the self
 * stream's buffer; open = party 0/1 between the rounds: ef = (X-A)|(Y-B); finish = after round 2:
 * Z = C + E*B + A*F + party*E*F, written in layoutZ */
int mpcx_snn_beaver_dealer_matmul(mpcx_ctx *ctx, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, int64_t *msg0,
                                  int64_t *msg1);
/* the dealer's triple for rows [row0, row0 + rows) of the product: the rows of A_i and C_i (rows x l, rows x c) and, when
 * B0 / B1 are non-NULL, all of B_i (l x c) -- the row-block form that spreads the dealer's product over GPUs */
int mpcx_snn_beaver_dealer_matmul_rows(mpcx_ctx *ctx, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, uint64_t row0,
                                       uint64_t rows, int64_t *A0, int64_t *A1, int64_t *B0, int64_t *B1, int64_t *C0,
                                       int64_t *C1);
int mpcx_snn_beaver_party_open_matmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, int layout,
                                      const int64_t *triple, uint64_t r, uint64_t l, uint64_t c, int64_t *ef_out);
int mpcx_snn_beaver_party_finish_matmul(mpcx_ctx *ctx, int party, const int64_t *triple, const int64_t *ef_own,
                                        const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int layoutZ,
                                        int64_t *Z);
/* vector_mul (:171-316) */
int mpcx_snn_beaver_dealer_vecmul(mpcx_ctx *ctx, uint64_t seedS, uint64_t n_total, uint64_t i0, uint64_t n,
                                  int64_t *A0, int64_t *B0, int64_t *C0, int64_t *A1, int64_t *B1, int64_t *C1);
int mpcx_snn_beaver_party_open_vecmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, const int64_t *A,
                                      const int64_t *B, uint64_t n, int64_t *E, int64_t *F);
int mpcx_snn_beaver_party_finish_vecmul(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *B,
                                        const int64_t *C, const int64_t *E_own, const int64_t *F_own,
                                        const int64_t *E_peer, const int64_t *F_peer, uint64_t n, int64_t *Z);
/* privateCompare (:507-680).  party (0/1): d = s * c over Z_67 with s,u regenerated on the device from the
 * s01 buffer (:523-534); helper (party 2): beta'[i] = any_k((d0+d1)[i][k] % 67 == 0) (:655-669) */
int mpcx_snn_pc_party(mpcx_ctx *ctx, int party, const uint16_t *xbits, const int8_t *beta, const int64_t *r,
                      uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01, uint16_t *d_out);
int mpcx_snn_pc_helper(mpcx_ctx *ctx, const uint16_t *d0, const uint16_t *d1, uint64_t n, int8_t *beta_out);
/* shareConvert (:685-893) */
int mpcx_snn_sc_party_open(mpcx_ctx *ctx, int party, const int64_t *A, uint64_t n_total, uint64_t i0, uint64_t n,
                           uint64_t seed01, int64_t *x_out);
int mpcx_snn_sc_party_pc_inputs(mpcx_ctx *ctx, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                                int8_t *eta2_out, int64_t *rm1_out);
int mpcx_snn_sc_party_finish(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *delta_j,
                             const int64_t *eta1_j, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seed01,
                             int64_t *y_out);
int mpcx_snn_sc_helper_split(mpcx_ctx *ctx, const int64_t *x0, const int64_t *x1, uint64_t n_total, uint64_t i0,
                             uint64_t n, uint64_t seedS, uint16_t *xbits0, int64_t *delta0, uint16_t *xbits1,
                             int64_t *delta1);
int mpcx_snn_sc_helper_eta(mpcx_ctx *ctx, const int8_t *eta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                           uint64_t seedS, int64_t *eta0, int64_t *eta1);
/* msb_L_1 (:898-1084) */
int mpcx_snn_msbl1_helper_deal(mpcx_ctx *ctx, uint64_t n_total, uint64_t i0, uint64_t n, uint64_t seedS, int64_t *x0,
                               int64_t *xlsb0, uint16_t *xbits0, int64_t *x1, int64_t *xlsb1, uint16_t *xbits1);
int mpcx_snn_msbl1_helper_beta(mpcx_ctx *ctx, const int8_t *beta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                               uint64_t seedS, int64_t *b0, int64_t *b1);
int mpcx_snn_msbl1_party_open(mpcx_ctx *ctx, const int64_t *A, const int64_t *xj, uint64_t n, int64_t *r_out);
int mpcx_snn_msbl1_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0,
                                   uint64_t n, uint64_t seed01, int8_t *beta_out, int64_t *R_out);
int mpcx_snn_msbl1_party_mid(mpcx_ctx *ctx, int party, const int64_t *beta1_j, const int64_t *xlsb_j,
                             const int8_t *beta0, const int64_t *R, uint64_t n, int64_t *gamma_out,
                             int64_t *delta_out);
int mpcx_snn_msbl1_party_finish(mpcx_ctx *ctx, const int64_t *gamma, const int64_t *delta, const int64_t *theta,
                                uint64_t n, int64_t *B);

/* msb (:1089-1101 = shareConvert of 2a, then msb_L_1) for parties 0 / 1 with ONE call per gap between two messenger
 * rounds (a party that sends nothing in a round may do the work on either side of it):
 *  party_open          before shareConvert's round 1: x_j = 2 a_j + r_j                (= ring_shl1 + sc_party_open)
 *  party_convert_open  after msb_L_1's round 1 (x_j of the deal has arrived): the end of shareConvert on 2 a_j, then
 *                      r_j = 2 y_j +L x_j                                      (= sc_party_finish + msbl1_party_open)
 *  party_pc_inputs     after msb_L_1's round 2, in one pass                            (= msbl1_party_pc_inputs)
 *  party_mid_open      after vector_mul's round 1 (A, B of the triple have arrived): gamma_j, delta_j and the masked
 *                      operands E, F                        (= msbl1_party_mid + beaver_party_open_vecmul)
 *  party_finish        after vector_mul's round 2       (= beaver_party_finish_vecmul + msbl1_party_finish)
 * A is the msb() input a_j itself (not doubled).  gamma_out / delta_out of party_mid_open and gamma / delta of
 * party_finish may both be NULL: party_finish then forms gamma + delta again as (E_own + A) + (F_own + B), which saves
 * writing and reading the two arrays. */
int mpcx_snn_msb_party_open(mpcx_ctx *ctx, int party, const int64_t *A, uint64_t n_total, uint64_t i0, uint64_t n,
                            uint64_t seed01, int64_t *x_out);
int mpcx_snn_msb_party_convert_open(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *delta_j,
                                    const int64_t *eta1_j, const int64_t *xj, uint64_t n_total, uint64_t i0, uint64_t n,
                                    uint64_t seed01, int64_t *r_out);
int mpcx_snn_msb_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0, uint64_t n,
                                 uint64_t seed01, int8_t *beta_out, int64_t *R_out);
int mpcx_snn_msb_party_mid_open(mpcx_ctx *ctx, int party, const int64_t *beta1_j, const int64_t *xlsb_j,
                                const int8_t *beta0, const int64_t *R, const int64_t *A, const int64_t *B, uint64_t n,
                                int64_t *gamma_out, int64_t *delta_out, int64_t *E, int64_t *F);
int mpcx_snn_msb_party_finish(mpcx_ctx *ctx, int party, const int64_t *A, const int64_t *B, const int64_t *C,
                              const int64_t *E_own, const int64_t *F_own, const int64_t *E_peer, const int64_t *F_peer,
                              const int64_t *gamma, const int64_t *delta, uint64_t n, int64_t *Bout);

/* ---- MyEngine (mpc_engine/MyEngine.cpp): additive shares on parties 0/1, party 2 deals ----
 * seed0 / seed1 = the streams party 2 shares with party 0 / 1 (its prgs[0] / prgs[1], KEY20 / KEY12, :21-24);
 * seed2 = that same stream as seen from party 0 / 1 (its prgs[2]); seedS = party 2's self stream; seed01 = the
 * stream parties 0 and 1 share.  All randomness here is read from the generators' buffers (int_buff / bool_buff
 * at size-dependent offsets), never popped.
 * share / share_matrix (:54-89) = mpcx_snn_share_local; reveal / reveal_matrix (:91-143) = mpcx_ring_add;
 * reveal_L_1 (:809-839) = mpcx_snn_add_L1; reveal_zp (:841-877) = mpcx_snn_add_zp; privateCompare's party 2
 * (:402-426) = mpcx_snn_pc_helper. */
/* matrix_mul (:145-219), one round.  dealer: C_j (row-major) is the message to party j.  open: ef = (X-A)|(Y-B)
 * row-major, the message to the peer.  finish: Z = C + E*B + A*F + party*E*F in layoutZ. */
int mpcx_my_beaver_dealer_matmul(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r, uint64_t l,
                                 uint64_t c, int64_t *C0, int64_t *C1);
int mpcx_my_beaver_party_open_matmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, int layout, uint64_t seed2,
                                     uint64_t r, uint64_t l, uint64_t c, int64_t *ef_out);
int mpcx_my_beaver_party_finish_matmul(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                       const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int layoutZ,
                                       int64_t *Z);
/* the row-block forms: rows [row0, row0 + rows) of the r x l . l x c product.  dealer: C0 / C1 hold rows x c words.
 * finish: C holds rows x c words, ef_own / ef_peer the rows of E (rows x l) followed by all of F (l x c); A and B are read
 * from the stream at their places in the whole product; Z (rows x c) in layoutZ. */
int mpcx_my_beaver_dealer_matmul_rows(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r, uint64_t l,
                                      uint64_t c, uint64_t row0, uint64_t rows, int64_t *C0, int64_t *C1);
int mpcx_my_beaver_party_finish_matmul_rows(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                            const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, uint64_t row0,
                                            uint64_t rows, int layoutZ, int64_t *Z);
/* vector_mul (:221-280), one round; ef = E[n] | F[n] */
int mpcx_my_beaver_dealer_vecmul(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t n_total,
                                 uint64_t i0, uint64_t n, int64_t *C0, int64_t *C1);
int mpcx_my_beaver_party_open_vecmul(mpcx_ctx *ctx, const int64_t *X, const int64_t *Y, uint64_t seed2, uint64_t n_total,
                                     uint64_t i0, uint64_t n, int64_t *ef_out);
int mpcx_my_beaver_party_finish_vecmul(mpcx_ctx *ctx, int party, uint64_t seed2, const int64_t *C, const int64_t *ef_own,
                                       const int64_t *ef_peer, uint64_t n_total, uint64_t i0, uint64_t n, int64_t *Z);
/* privateCompare, party 0/1 (:293-357): as SecureNN's without the r = -1 branch (beta != 0 compares against r + 1) */
int mpcx_my_pc_party(mpcx_ctx *ctx, int party, const uint16_t *xbits, const int8_t *beta, const int64_t *r, uint64_t i0,
                     uint64_t n, uint64_t seed01, uint16_t *d_out);
/* msb (:442-597), three rounds, the result is an XOR share (one byte per element).
 *  helper_deal (party 2, before round 1): message to party j = xlsb_j[n] | xbits_j[n][64]
 *  party_open (party 0/1, before round 1): r_j = (buf2[k] % L_1) +L (2a +L 2a), the message to the peer
 *  party_pc_inputs: the plaintext privateCompare inputs of round 2 (beta = boolbuf01[k], R = r_peer +L r_own)
 *  helper_beta (party 2, after round 2): XOR shares of beta', the round-3 messages
 *  party_finish (after round 3) */
int mpcx_my_msb_helper_deal(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t i0, uint64_t n,
                            int8_t *xlsb0, uint16_t *xbits0, int8_t *xlsb1, uint16_t *xbits1);
int mpcx_my_msb_helper_beta(mpcx_ctx *ctx, const int8_t *beta1_plain, uint64_t n_total, uint64_t i0, uint64_t n,
                            uint64_t seedS, int8_t *b0, int8_t *b1);
int mpcx_my_msb_party_open(mpcx_ctx *ctx, const int64_t *A, uint64_t seed2, uint64_t i0, uint64_t n, int64_t *r_out);
int mpcx_my_msb_party_pc_inputs(mpcx_ctx *ctx, const int64_t *r_own, const int64_t *r_peer, uint64_t i0, uint64_t n,
                                uint64_t seed01, int8_t *beta_out, int64_t *R_out);
int mpcx_my_msb_party_finish(mpcx_ctx *ctx, int party, const int8_t *beta1_j, const int8_t *beta0, const int8_t *xlsb_j,
                             const int64_t *R, uint64_t n, int8_t *B);
/* bit_inject (:599-772), one round.  dealer: message to party j = V_j|W_j|S_j|T_j.  pre (party 0/1): message to the
 * peer = y_j[n] | c_j[n bytes].  post: Z_j. */
int mpcx_my_bit_inject_dealer(mpcx_ctx *ctx, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t n_total, uint64_t i0,
                              uint64_t n, int64_t *V0, int64_t *W0, int64_t *S0, int64_t *T0, int64_t *V1, int64_t *W1,
                              int64_t *S1, int64_t *T1);
int mpcx_my_bit_inject_party_pre(mpcx_ctx *ctx, const int64_t *X, const int8_t *A, uint64_t seed2, uint64_t i0, uint64_t n,
                                 int64_t *y_out, int8_t *c_out);
int mpcx_my_bit_inject_party_post(mpcx_ctx *ctx, int party, uint64_t seed2, uint64_t i0, uint64_t n, const int64_t *y_own,
                                  const int8_t *c_own, const int64_t *y_peer, const int8_t *c_peer, const int64_t *V,
                                  const int64_t *W, const int64_t *S, const int64_t *T, int64_t *Z);
/* reveal_vec_bools (:777-807): out = a ^ b, one byte per boolean */
int mpcx_xor_bools(mpcx_ctx *ctx, const int8_t *a, const int8_t *b, int8_t *out, uint64_t n);

/* ---- several GPUs behind one party (no counterpart in the reference, which is single-threaded host code) ----
 * The path shards without any exchange between the devices: matrix_mul by row blocks of X (ABY3Engine.cpp:189-197:
 * Y replicated, the mask words of a block start row0 * c words into both pop streams), the element-wise segments by
 * contiguous element ranges (their (n_total, i0, n) arguments).  A group is one context per listed device (a device
 * may be listed more than once) and one host thread per member; every member moves its own slice between host
 * memory and its device, so the slices of a message land in the messenger-facing buffer directly. */
typedef struct mpcx_group mpcx_group;
int mpcx_group_init(const int *devices, int n, mpcx_group **out); /* error text: mpcx_last_error(NULL) */
int mpcx_group_shutdown(mpcx_group *g);
int mpcx_group_size(const mpcx_group *g);
mpcx_ctx *mpcx_group_ctx(mpcx_group *g, int member);
const char *mpcx_group_last_error(mpcx_group *g);
/* part `part` of `parts` contiguous parts of n items, in multiples of `granule` (the first parts are the larger ones) */
int mpcx_plan_split(uint64_t n, int parts, int part, uint64_t granule, uint64_t *first, uint64_t *count);
/* fn(ctx, member, first, count, user) for every member's range of [0, n), each on its member's thread with the
 * member's device current; returns the first non-zero result.  A vector shorter than one granule per member uses
 * fewer members. */
typedef int (*mpcx_member_fn)(mpcx_ctx *ctx, int member, uint64_t first, uint64_t count, void *user);
int mpcx_group_for_each(mpcx_group *g, uint64_t n, uint64_t granule, mpcx_member_fn fn, void *user);
/* mpcx_aby3_matmul_local over the group.  X0, X1, Y0, Y1 and Z0 are HOST pointers: every member uploads its rows of
 * X and all of Y, multiplies, and writes its rows of Z0 (row-major) into place. */
int mpcx_group_aby3_matmul_local(mpcx_group *g, const int64_t *X0, const int64_t *X1, const int64_t *Y0,
                                 const int64_t *Y1, uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next,
                                 uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0);
/* A replicated operand kept on the members' devices: every member uploads both shares of Y (rows x cols, `layout`) once
 * and keeps a row-major copy until mpcx_group_operand_free; mpcx_group_aby3_matmul_local_resident is
 * mpcx_group_aby3_matmul_local with that operand in place of the host pointers Y0 / Y1, so that a product moves only the
 * members' row blocks of X and Z (a weight matrix used by many products; for the 16384 x 16384 configuration Y is 4 GiB
 * per member and call otherwise).  No counterpart in the reference. */
typedef struct mpcx_group_operand mpcx_group_operand;
int mpcx_group_operand_upload(mpcx_group *g, const int64_t *M0, const int64_t *M1, uint64_t rows, uint64_t cols, int layout,
                              mpcx_group_operand **out);
int mpcx_group_operand_free(mpcx_group *g, mpcx_group_operand *op);
int mpcx_group_aby3_matmul_local_resident(mpcx_group *g, const int64_t *X0, const int64_t *X1, const mpcx_group_operand *Y,
                                          uint64_t r, uint64_t l, uint64_t c, int layout, uint64_t seed_next,
                                          uint64_t pop_next, uint64_t seed_prev, uint64_t pop_prev, int64_t *Z0);
/* mpcx_snn_beaver_party_finish_matmul (SecurennEngine.cpp:157-167) over the group.  triple (A | B | C), ef_own and ef_peer
 * (E | F) are HOST pointers in the wire order of the messages that carry them, Z (r x c, row-major) a host pointer:
 * every member uploads its rows of A, C and E and all of B and F, and writes its rows of Z into place. */
int mpcx_group_snn_beaver_party_finish_matmul(mpcx_group *g, int party, const int64_t *triple, const int64_t *ef_own,
                                              const int64_t *ef_peer, uint64_t r, uint64_t l, uint64_t c, int64_t *Z);
/* mpcx_snn_beaver_dealer_matmul (SecurennEngine.cpp:72-110) over the group: msg0, msg1 (A_i | B_i | C_i, r l + l c + r c
 * words each) are HOST pointers; every member deals the rows of A_i and C_i of its row block into place, the first
 * member also B_i. */
int mpcx_group_snn_beaver_dealer_matmul(mpcx_group *g, uint64_t seedS, uint64_t r, uint64_t l, uint64_t c, int64_t *msg0,
                                        int64_t *msg1);
/* MyEngine::matrix_mul (MyEngine.cpp:145-219) over the group, host pointers throughout: the dealer's C_0 / C_1 (r x c,
 * row-major = the messages), and the parties' Z (r x c, row-major) from C, ef_own and ef_peer (E | F) as the messages
 * carry them. */
int mpcx_group_my_beaver_dealer_matmul(mpcx_group *g, uint64_t seed0, uint64_t seed1, uint64_t seedS, uint64_t r,
                                       uint64_t l, uint64_t c, int64_t *C0, int64_t *C1);
int mpcx_group_my_beaver_party_finish_matmul(mpcx_group *g, int party, uint64_t seed2, const int64_t *C,
                                             const int64_t *ef_own, const int64_t *ef_peer, uint64_t r, uint64_t l,
                                             uint64_t c, int64_t *Z);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
