// SecureNN engine: parties 0 and 1 hold additive shares over Z_2^64, party 2 assists (deals Beaver triples,
// runs the comparison's third role).
//
// Source-compatible with callers of the reference class (mpc_engine/SecurennEngine.hpp:20-50): constructor,
// public data members and every method keep their names, argument types and out-parameter style.  Rounds and
// wire format are the reference's; the arithmetic between rounds runs on the GPU through the C ABI of
// include/mpcx.h.  Randomness that the reference reads from PRgenerator::int_buff / bool_buff at size-dependent
// offsets is regenerated on the device for every index (the reference's own buffers stop at BUFF_NUM = 160000
// words), so vectors and matrices of any size work and agree with the reference wherever the reference is
// defined.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "../eigen-3.4.0/Eigen/Dense"
#include "../utils/PRgenerator.hpp"
#include "../utils/VirtuleMessenger.hpp"

#define KEY01 0xAB01
#define KEY12 0xBC12
#define KEY20 0xCA20
#define KEY0 0xF0
#define KEY1 0xF0
#define KEY2 0xF0

using std::string;
using std::vector;
using MatrixXl = Eigen::Matrix<int64_t, Eigen::Dynamic, Eigen::Dynamic>;
using zpshare = std::array<uint16_t, 64>; // Z_67 shares of the 64 bits of one element, least significant first

struct mpcx_ctx;
struct mpcx_group;

class SecurennEngine {
  public:
    using Words = vector<int64_t>;
    using Bits = vector<int8_t>;

    SecurennEngine(VirtuleMessenger *_messenger, size_t _party_id);
    SecurennEngine(const SecurennEngine &other) = delete;
    ~SecurennEngine();

    // ---- state the reference exposes
    VirtuleMessenger *messenger; // borrowed
    size_t party_id;
    vector<PRgenerator> prgs;    // prgs[i]: the stream shared with party i; prgs[party_id]: this party's own
    vector<string> message_send;
    vector<string> message_recv;
    mpcx_ctx *gpu;               // this engine's GPU context (an addition)
    mpcx_group *group;           // MPCX_DEVICES=a,b,..: the element-wise protocols spread over these GPUs (else NULL)

    // ---- input / output (SecurennEngine.cpp:318-502)
    void share(const Words &plain, Words &mine);
    void reveal(const Words &mine, Words &plain);
    void reveal_matrix(const MatrixXl &mine, MatrixXl &plain);
    void reveal_vec_bools(const Bits &mine, Bits &plain);
    void reveal_L_1(const Words &mine, Words &plain);                           // shares over Z_(2^64 - 1)
    void reveal_zp(const vector<zpshare> &mine, vector<zpshare> &plain);        // shares over Z_67

    // ---- Beaver products, party 2 deals (:61-316)
    void matrix_mul(const MatrixXl &X, const MatrixXl &Y, MatrixXl &Z);
    void vector_mul(const Words &X, const Words &Y, Words &Z);

    // ---- comparison and sign extraction (:507-1112)
    void privateCompare(const vector<zpshare> &xbits, const Bits &beta, const Words &r, Bits &beta_prime);
    void shareConvert(const Words &A, Words &Y);
    void msb_L_1(const Words &A, Words &B);
    void msb(const Words &A, Words &B);

    // ---- additions (no counterpart in the reference class; everything above is unchanged)
    // R = ReLU(A) = A - vector_mul(A, msb(A)), the composition the reference benchmarks (benchmark/benchmark1.cpp:135-136)
    // followed by the subtraction, as ONE protocol: the same twelve rounds and messages, with the sign shares, the
    // product and the input kept on the device in between.  Bit-identical to the three steps done one after the other.
    void relu(const Words &A, Words &R);
    // Keeps a device copy of a share matrix (e.g. a weight matrix) until unpin: matrix_mul finds an operand by its
    // address and skips the upload.  The caller promises not to change or destroy a pinned matrix.
    void pin(const MatrixXl &M);
    void unpin(const MatrixXl &M);

  private:
    struct Pinned;
    std::vector<Pinned *> pinned;
    Pinned *find_pinned(const MatrixXl &M) const;
};
