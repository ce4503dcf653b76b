// MyEngine: parties 0 and 1 hold additive shares over Z_2^64; party 2 deals correlated randomness out of the
// buffers of the pairwise generators and plays the third role of the comparison.
//
// Source-compatible with callers of the reference class (mpc_engine/MyEngine.hpp:24-48): constructor, public
// data members and every method keep their names, argument types and out-parameter style.  Rounds and wire
// format are the reference's; the arithmetic between rounds runs on the GPU through the C ABI of
// include/mpcx.h.  Buffer words are regenerated on the device for every index, so sizes beyond
// BUFF_NUM = 160000 words work.
#pragma once
#include <array>
#include <bitset>
#include <string>
#include <vector>

#include "../eigen-3.4.0/Eigen/Dense"
#include "../utils/PRgenerator.hpp"
#include "../utils/VirtuleMessenger.hpp"
#include "../utils/util.hpp"

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

class MyEngine {
  public:
    MyEngine(VirtuleMessenger *_messenger, size_t _party_id);
    MyEngine(const MyEngine &other) = delete;
    ~MyEngine();

    // ---- state the reference exposes
    VirtuleMessenger *messenger; // borrowed
    size_t party_id;
    vector<PRgenerator> prgs;    // prgs[i]: the stream shared with party i; prgs[party_id]: this party's own
    vector<string> message_send;
    vector<string> message_recv;
    mpcx_ctx *gpu;               // this engine's GPU context (an addition)
    mpcx_group *group;           // MPCX_DEVICES=a,b,..: the element-wise protocols spread over these GPUs (else NULL)

    // ---- input / output (MyEngine.cpp:54-143, :777-end)
    void share(const VecI64 &plain, VecI64 &mine);
    void share_matrix(const MatrixXl &plain, MatrixXl &mine);
    void reveal(const VecI64 &mine, VecI64 &plain);
    void reveal_matrix(const MatrixXl &mine, MatrixXl &plain);
    void reveal_vec_bools(const vecbool &mine, vecbool &plain);
    void reveal_L_1(const VecI64 &mine, VecI64 &plain);
    void reveal_zp(const vector<zpshare> &mine, vector<zpshare> &plain);

    // ---- Beaver products; party 2 reads A and B out of the buffers of prgs[0] / prgs[1] (:145-285)
    void matrix_mul(const MatrixXl &X, const MatrixXl &Y, MatrixXl &Z);
    void vector_mul(const VecI64 &X, const VecI64 &Y, VecI64 &Z);

    // ---- comparison, sign bit as XOR-shared bits, one-round arithmetic-times-bit (:287-772)
    void privateCompare(const vector<zpshare> &xbits, const vecbool &beta, const VecI64 &r, vecbool &beta_prime);
    void msb(const VecI64 &A, vecbool &B);
    void bit_inject(const VecI64 &X, const vecbool &A, VecI64 &Z);
};
