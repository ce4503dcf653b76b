// ABY3 engine: three parties, replicated 2-out-of-3 sharing over Z_2^64 (party j holds shares j and j+1).
//
// Source-compatible with callers of the reference class (mpc_engine/ABY3Engine.hpp:24-54): constructor, public
// data members and every method keep their names, argument types and out-parameter style, so the reference's
// test and benchmark programs compile against this header unchanged.  Each method keeps the reference's
// sequence of messenger rounds and its wire format; what happens between two rounds runs on the GPU through
// the C ABI of include/mpcx.h, one call per local segment.
#pragma once
#include <array>
#include <bitset>
#include <string>
#include <vector>

#include "../eigen-3.4.0/Eigen/Dense"
#include "../utils/PRgenerator.hpp"
#include "../utils/VirtuleMessenger.hpp"
#include "../utils/util.hpp"

// generator keys: one per pair of parties, and each party's own (ABY3Engine.cpp:34-53)
#define KEY01 0xAB01
#define KEY12 0xBC12
#define KEY20 0xCA20
#define KEY0 0xF0
#define KEY1 0xF0
#define KEY2 0xF0

using std::array;
using std::bitset;
using std::string;
using std::vector;

struct mpcx_ctx;
struct mpcx_group;

class ABY3Engine {
  public:
    // the two shares a party holds of a vector / matrix / bit vector
    using VecPair = array<VecI64, 2>;
    using MatPair = array<MatrixXl, 2>;
    using BitPair = array<vecbool, 2>;

    ABY3Engine(VirtuleMessenger *_messenger, size_t _party_id);
    ABY3Engine(const ABY3Engine &other) = delete;
    ~ABY3Engine();

    // ---- state the reference exposes (its tests read prgs and party_id directly)
    VirtuleMessenger *messenger; // borrowed
    size_t party_id;
    vector<PRgenerator> prgs;    // prgs[i]: the stream shared with party i; prgs[party_id]: this party's own
    vector<string> message_send; // three slots each, reused from call to call
    vector<string> message_recv;
    mpcx_ctx *gpu;               // this engine's GPU context (an addition)
    // With MPCX_DEVICES=a,b,.. in the environment the engine spreads matrix_mul (row blocks) and add_prepare / msb /
    // bit_inject (element ranges) over those GPUs; gpu is then the first member's context.  NULL otherwise.
    mpcx_group *group;

    // ---- input: party 0 contributes the plain values, everyone ends up with its two shares (:63-127)
    void share(const VecI64 &plain, VecPair &out);
    void share_matrix(const MatrixXl &plain, MatPair &out);
    void share_bools(const vecbool &plain, BitPair &out);

    // ---- output: every party learns the plain values (:132-175, :277-289)
    void reveal(const VecPair &in, VecI64 &plain);
    void reveal_matrix(const MatPair &in, MatrixXl &plain);
    void reveal_bools(const BitPair &in, vecbool &plain);
    void reveal_bits(const VecPair &in, VecI64 &plain); // XOR-shared words

    // ---- products, one round each (:177-272)
    void matrix_mul(const MatPair &X, const MatPair &Y, MatPair &Z);
    void vector_mul(const VecPair &X, const VecPair &Y, VecPair &Z);
    void vecbool_mul(const BitPair &X, const BitPair &Y, BitPair &Z);

    // ---- sign extraction: carry generate / propagate words, then the prefix circuit (:295-467)
    void add_prepare(const VecPair &X, VecPair &G, VecPair &P);
    void msb(const VecPair &G, const VecPair &P, BitPair &Z);

    // ---- fixed-point rescale and arithmetic-times-bit (:469-1038)
    void trunc(const VecPair &X, VecPair &Y, size_t number);
    void bit_inject(const VecPair &A, const BitPair &B, VecPair &C);

    // ---- additions (no counterpart in the reference class; everything above is unchanged)
    // R = ReLU(X) = X - bit_inject(X, msb(add_prepare(X))), the composition the reference benchmarks
    // (benchmark/benchmark1.cpp:71-73) followed by the subtraction of the two share vectors, as ONE protocol: the same
    // nine rounds, messages and generator words as the three calls, but G, P, the carry tree and the sign bits stay
    // on the device between the rounds (only messages cross to the host).  Bit-identical to the composition.
    void relu(const VecPair &X, VecPair &R);
    // Keeps a device copy of a share pair (e.g. a weight matrix that many products use) until unpin: matrix_mul finds
    // an operand by the address of its pair and skips the upload.  Explicit on purpose: the caller promises not to
    // change or destroy a pinned pair.  The destructor releases what is still pinned.  With MPCX_DEVICES every GPU of
    // the group keeps a copy, and matrix_mul uses it for its second (replicated) operand.
    void pin(const MatPair &M);
    void unpin(const MatPair &M);

  private:
    struct Pinned;
    std::vector<Pinned *> pinned;
    Pinned *find_pinned(const MatPair &M) const;
    // a row-major message of rows x cols words into a matrix (large ones are turned on the device)
    void wire_to_matrix(const string &wire, size_t rows, size_t cols, MatrixXl &M);
};
