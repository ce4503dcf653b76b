// SecureNN engine with the reference's public surface (mpc_engine/SecurennEngine.hpp:20-50): same
// constructor, public members and method signatures; rounds and wire format as in the reference, the
// arithmetic between rounds on the GPU through the C ABI in include/mpcx.h.  Randomness that the
// reference reads from PRgenerator::int_buff / bool_buff at size-dependent offsets is regenerated on
// the device for every index (the reference's own buffers stop at BUFF_NUM = 160000 words), so vectors
// and matrices of any size work and agree with the reference wherever the reference is defined.
#pragma once
#include "../utils/VirtuleMessenger.hpp"
#include "../utils/PRgenerator.hpp"
#include "../eigen-3.4.0/Eigen/Dense"
#include <array>
#include <string>
#include <vector>
#define KEY01 0xAB01
#define KEY12 0xBC12
#define KEY20 0xCA20
#define KEY0 0xF0
#define KEY1 0xF0
#define KEY2 0xF0
using MatrixXl = Eigen::Matrix<int64_t, Eigen::Dynamic, Eigen::Dynamic>;
// 64 shares over Z_67 of the bits of one element, least significant bit first
using zpshare = std::array<uint16_t, 64>;
using std::string;
using std::vector;
struct mpcx_ctx;

class SecurennEngine {
  public:
    VirtuleMessenger *messenger;
    size_t party_id;
    vector<PRgenerator> prgs;
    vector<string> message_send;
    vector<string> message_recv;
    SecurennEngine(VirtuleMessenger *_messenger, size_t _party_id);
    SecurennEngine(const SecurennEngine &other) = delete;
    void matrix_mul(const MatrixXl &Xj, const MatrixXl &Yj, MatrixXl &Zj);
    void vector_mul(const vector<int64_t> &Xj, const vector<int64_t> &Yj, vector<int64_t> &Zj);
    void reveal(const vector<int64_t> &Xj, vector<int64_t> &X_plain);
    void reveal_matrix(const MatrixXl &Xj, MatrixXl &X_plain);
    void reveal_vec_bools(const vector<int8_t> &Xj, vector<int8_t> &X_plain);

    void reveal_L_1(const vector<int64_t> &Xj, vector<int64_t> &X_plain);
    void reveal_zp(const vector<zpshare> &Xbits_j, vector<zpshare> &Xbits_plain);

    void share(const vector<int64_t> &X_plain, vector<int64_t> &Xj);
    void privateCompare(const vector<zpshare> &Xbits_j, const vector<int8_t> &beta1_plain,
                        const vector<int64_t> &r_palin, vector<int8_t> &beta2_plain);
    void shareConvert(const vector<int64_t> &Aj, vector<int64_t> &Yj);
    void msb_L_1(const vector<int64_t> &Aj, vector<int64_t> &Bj);
    void msb(const vector<int64_t> &Aj, vector<int64_t> &Bj);

    ~SecurennEngine();

    mpcx_ctx *gpu; // the engine's GPU context
};
