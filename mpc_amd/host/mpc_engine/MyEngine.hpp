// MyEngine with the reference's public surface (mpc_engine/MyEngine.hpp:24-48): same constructor, public
// members and method signatures.  Two-party additive shares on parties 0/1, party 2 deals the correlated
// randomness out of the pairwise generators' buffers; every method keeps the reference's rounds and wire format,
// the arithmetic between rounds runs on the GPU through the C ABI in include/mpcx.h.  Buffer words are
// regenerated on the device for every index, so sizes beyond BUFF_NUM = 160000 words work.
#pragma once
#include "../utils/VirtuleMessenger.hpp"
#include "../utils/PRgenerator.hpp"
#include "../eigen-3.4.0/Eigen/Dense"
#include "../utils/util.hpp"
#include <array>
#include <bitset>
#include <vector>

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
class MyEngine {
  public:
    VirtuleMessenger *messenger;
    size_t party_id;
    vector<PRgenerator> prgs;
    vector<string> message_send;
    vector<string> message_recv;
    MyEngine(VirtuleMessenger *_messenger, size_t _party_id);
    ~MyEngine();
    MyEngine(const MyEngine &other) = delete;
    void share(const VecI64 &X_plain, VecI64 &Xj);
    void share_matrix(const MatrixXl &X_plain, MatrixXl &Xj);
    void reveal(const VecI64 &Xj, VecI64 &X_plain);
    void reveal_matrix(const MatrixXl &Xj, MatrixXl &X_plain);
    void matrix_mul(const MatrixXl &Xj, const MatrixXl &Yj, MatrixXl &Zj);
    void vector_mul(const VecI64 &Xj, const VecI64 &Yj, VecI64 &Zj);
    void privateCompare(const vector<zpshare> &Xbits_j, const vecbool &beta1_plain, const VecI64 &r_palin,
                        vecbool &beta2_plain);
    void msb(const VecI64 &Aj, vecbool &Bj);
    void bit_inject(const VecI64 &Xj, const vecbool &Aj, VecI64 &Zj);
    void reveal_vec_bools(const vecbool &Xj, vecbool &X_plain);
    void reveal_L_1(const VecI64 &Xj, VecI64 &X_plain);
    void reveal_zp(const vector<zpshare> &Xbits_j, vector<zpshare> &Xbits_plain);

    mpcx_ctx *gpu; // the engine's GPU context
};
