// Compile-time proof of the drop-in surface: every public method of the reference's engine, generator and messenger
// classes exists here with exactly the reference's signature, every public data member with exactly its type, and the
// constants have their values.  Nothing runs: the file is compiled with -fsyntax-only (tests/test_host_engines.py).
// The statements are written from the reference's declarations (mpc_engine/ABY3Engine.hpp:24-54,
// mpc_engine/SecurennEngine.hpp:20-50, mpc_engine/MyEngine.hpp:24-48, utils/PRgenerator.hpp:6-24,
// utils/SocketMessenger.hpp:12-26, utils/VirtuleMessenger.hpp:6-16, utils/util.hpp:7-12) and from how its callers use
// them (test/test_securenn.cpp:27,90,156 and test/test_ABY3.cpp:256-276 reach into prgs[k].pop_int64(), int_buff,
// party_id directly).
#include <type_traits>

#include "../mpc_engine/ABY3Engine.hpp"
#include "../mpc_engine/MyEngine.hpp"
#include "../mpc_engine/SecurennEngine.hpp"
#include "../utils/PRgenerator.hpp"
#include "../utils/SocketMessenger.hpp"
#include "../utils/VirtuleMessenger.hpp"
#include "../utils/util.hpp"

#define SAME(expr, ...) static_assert(std::is_same<decltype(expr), __VA_ARGS__>::value, #expr)

// ---- types and constants (utils/util.hpp:7-12)
static_assert(std::is_same<MatrixXl, Eigen::Matrix<int64_t, Eigen::Dynamic, Eigen::Dynamic>>::value, "MatrixXl");
static_assert(std::is_same<VecI64, Eigen::Array<int64_t, Eigen::Dynamic, 1>>::value, "VecI64");
static_assert(std::is_same<zpshare, std::array<uint16_t, 64>>::value, "zpshare");
static_assert(std::is_same<vecbool, std::vector<int8_t>>::value, "vecbool");
static_assert(P == 67, "P");
static_assert((uint64_t)L_1 == 0xFFFFFFFFFFFFFFFFull, "L_1");
static_assert(BUFF_NUM == 160000, "BUFF_NUM");
static_assert(PORT == 8000, "PORT");
static_assert(KEY01 == 0xAB01 && KEY12 == 0xBC12 && KEY20 == 0xCA20 && KEY0 == 0xF0 && KEY1 == 0xF0 && KEY2 == 0xF0, "keys");

// ---- PRgenerator (utils/PRgenerator.hpp:7-24)
SAME(&PRgenerator::seed, uint64_t PRgenerator::*);
SAME(&PRgenerator::lcg, uint64_t PRgenerator::*);
SAME(&PRgenerator::hash, uint64_t PRgenerator::*);
SAME(&PRgenerator::int_buff, std::vector<int64_t> PRgenerator::*);
SAME(&PRgenerator::bool_buff, std::vector<int8_t> PRgenerator::*);
SAME(&PRgenerator::pop_int64, int64_t (PRgenerator::*)());
SAME(&PRgenerator::pop_bool, int8_t (PRgenerator::*)());
static_assert(std::is_constructible<PRgenerator, uint64_t, uint64_t>::value && !std::is_default_constructible<PRgenerator>::value,
              "PRgenerator(seed, lcg) only");

// ---- messengers (utils/VirtuleMessenger.hpp:6-16, utils/SocketMessenger.hpp:14-26)
SAME(&VirtuleMessenger::send_and_recv, void (VirtuleMessenger::*)(const std::vector<std::string> &, std::vector<std::string> &));
static_assert(std::is_abstract<VirtuleMessenger>::value && std::has_virtual_destructor<VirtuleMessenger>::value, "VirtuleMessenger");
static_assert(std::is_base_of<VirtuleMessenger, SocketMessenger>::value && std::is_constructible<SocketMessenger, size_t>::value &&
                  !std::is_copy_constructible<SocketMessenger>::value,
              "SocketMessenger");
SAME(&SocketMessenger::party_id, size_t SocketMessenger::*);
SAME(&SocketMessenger::server_fd, int SocketMessenger::*);
SAME(&SocketMessenger::recv_fd, int (SocketMessenger::*)[3]);
SAME(&SocketMessenger::send_fd, int (SocketMessenger::*)[3]);

// ---- the state every engine exposes
template <class E> struct EngineState {
    SAME(&E::messenger, VirtuleMessenger *E::*);
    SAME(&E::party_id, size_t E::*);
    SAME(&E::prgs, std::vector<PRgenerator> E::*);
    SAME(&E::message_send, std::vector<std::string> E::*);
    SAME(&E::message_recv, std::vector<std::string> E::*);
    static_assert(std::is_constructible<E, VirtuleMessenger *, size_t>::value, "Engine(VirtuleMessenger *, size_t)");
    static constexpr bool ok = true;
};
static_assert(EngineState<ABY3Engine>::ok && EngineState<SecurennEngine>::ok && EngineState<MyEngine>::ok, "engine state");
static_assert(!std::is_copy_constructible<ABY3Engine>::value && !std::is_copy_constructible<SecurennEngine>::value, "no copies");

// ---- ABY3Engine (mpc_engine/ABY3Engine.hpp:34-50)
using V2 = std::array<VecI64, 2>;
using M2 = std::array<MatrixXl, 2>;
using B2 = std::array<vecbool, 2>;
SAME(&ABY3Engine::share, void (ABY3Engine::*)(const VecI64 &, V2 &));
SAME(&ABY3Engine::share_matrix, void (ABY3Engine::*)(const MatrixXl &, M2 &));
SAME(&ABY3Engine::share_bools, void (ABY3Engine::*)(const vecbool &, B2 &));
SAME(&ABY3Engine::reveal, void (ABY3Engine::*)(const V2 &, VecI64 &));
SAME(&ABY3Engine::reveal_matrix, void (ABY3Engine::*)(const M2 &, MatrixXl &));
SAME(&ABY3Engine::reveal_bools, void (ABY3Engine::*)(const B2 &, vecbool &));
SAME(&ABY3Engine::reveal_bits, void (ABY3Engine::*)(const V2 &, VecI64 &));
SAME(&ABY3Engine::matrix_mul, void (ABY3Engine::*)(const M2 &, const M2 &, M2 &));
SAME(&ABY3Engine::vector_mul, void (ABY3Engine::*)(const V2 &, const V2 &, V2 &));
SAME(&ABY3Engine::vecbool_mul, void (ABY3Engine::*)(const B2 &, const B2 &, B2 &));
SAME(&ABY3Engine::add_prepare, void (ABY3Engine::*)(const V2 &, V2 &, V2 &));
SAME(&ABY3Engine::msb, void (ABY3Engine::*)(const V2 &, const V2 &, B2 &));
SAME(&ABY3Engine::trunc, void (ABY3Engine::*)(const V2 &, V2 &, size_t));
SAME(&ABY3Engine::bit_inject, void (ABY3Engine::*)(const V2 &, const B2 &, V2 &));

// ---- SecurennEngine (mpc_engine/SecurennEngine.hpp:31-46)
using W = std::vector<int64_t>;
using Bv = std::vector<int8_t>;
using Zp = std::vector<zpshare>;
SAME(&SecurennEngine::matrix_mul, void (SecurennEngine::*)(const MatrixXl &, const MatrixXl &, MatrixXl &));
SAME(&SecurennEngine::vector_mul, void (SecurennEngine::*)(const W &, const W &, W &));
SAME(&SecurennEngine::reveal, void (SecurennEngine::*)(const W &, W &));
SAME(&SecurennEngine::reveal_matrix, void (SecurennEngine::*)(const MatrixXl &, MatrixXl &));
SAME(&SecurennEngine::reveal_vec_bools, void (SecurennEngine::*)(const Bv &, Bv &));
SAME(&SecurennEngine::reveal_L_1, void (SecurennEngine::*)(const W &, W &));
SAME(&SecurennEngine::reveal_zp, void (SecurennEngine::*)(const Zp &, Zp &));
SAME(&SecurennEngine::share, void (SecurennEngine::*)(const W &, W &));
SAME(&SecurennEngine::privateCompare, void (SecurennEngine::*)(const Zp &, const Bv &, const W &, Bv &));
SAME(&SecurennEngine::shareConvert, void (SecurennEngine::*)(const W &, W &));
SAME(&SecurennEngine::msb_L_1, void (SecurennEngine::*)(const W &, W &));
SAME(&SecurennEngine::msb, void (SecurennEngine::*)(const W &, W &));

// ---- MyEngine (mpc_engine/MyEngine.hpp:34-47)
SAME(&MyEngine::share, void (MyEngine::*)(const VecI64 &, VecI64 &));
SAME(&MyEngine::share_matrix, void (MyEngine::*)(const MatrixXl &, MatrixXl &));
SAME(&MyEngine::reveal, void (MyEngine::*)(const VecI64 &, VecI64 &));
SAME(&MyEngine::reveal_matrix, void (MyEngine::*)(const MatrixXl &, MatrixXl &));
SAME(&MyEngine::matrix_mul, void (MyEngine::*)(const MatrixXl &, const MatrixXl &, MatrixXl &));
SAME(&MyEngine::vector_mul, void (MyEngine::*)(const VecI64 &, const VecI64 &, VecI64 &));
SAME(&MyEngine::privateCompare, void (MyEngine::*)(const Zp &, const vecbool &, const VecI64 &, vecbool &));
SAME(&MyEngine::msb, void (MyEngine::*)(const VecI64 &, vecbool &));
SAME(&MyEngine::bit_inject, void (MyEngine::*)(const VecI64 &, const vecbool &, VecI64 &));
SAME(&MyEngine::reveal_vec_bools, void (MyEngine::*)(const vecbool &, vecbool &));
SAME(&MyEngine::reveal_L_1, void (MyEngine::*)(const VecI64 &, VecI64 &));
SAME(&MyEngine::reveal_zp, void (MyEngine::*)(const Zp &, Zp &));

// ---- the codecs callers use (utils/util.hpp:19-105, :135-154, :184-200)
SAME(&matrix2string, void (*)(const MatrixXl &, std::string &));
SAME(&string2matrix, void (*)(const std::string &, MatrixXl &));
SAME(&vector2string, void (*)(const std::vector<int64_t> &, std::string &));
SAME(&string2vector, void (*)(const std::string &, std::vector<int64_t> &));
SAME(&vecI642string, void (*)(const VecI64 &, std::string &));
SAME(&string2vecI64, void (*)(const std::string &, VecI64 &));
SAME(&vecZpshare2string, void (*)(const std::vector<zpshare> &, std::string &));
SAME(&string2vecZpshare, void (*)(const std::string &, std::vector<zpshare> &));
SAME(&vecbool2string, void (*)(const std::vector<int8_t> &, std::string &));
SAME(&string2vecbool, void (*)(const std::string &, std::vector<int8_t> &));
SAME(&matrix2vecI64, void (*)(const MatrixXl &, VecI64 &));
SAME(&vecI642matrix, void (*)(const VecI64 &, MatrixXl &));
SAME(&matrix2vector, void (*)(const MatrixXl &, std::vector<int64_t> &));
SAME(&vector2matrix, void (*)(const std::vector<int64_t> &, MatrixXl &));

// how the reference's drivers reach into an engine (test/test_securenn.cpp:27,90,156; test/test_ABY3.cpp:256-276)
inline void caller_idioms(SocketMessenger &socketMessenger) {
    ABY3Engine aby3(&socketMessenger, socketMessenger.party_id);
    SecurennEngine snn(&socketMessenger, socketMessenger.party_id);
    int64_t a = snn.prgs[2].pop_int64() + aby3.prgs[aby3.party_id].int_buff[0];
    int8_t b = snn.prgs[0].pop_bool() + snn.prgs[1].bool_buff[0];
    (void)a, (void)b;
    MatrixXl X(3, 4);
    X(1, 2) = 7;
    VecI64 v(5);
    v[0] = v(1) = (int64_t)X.rows() * X.cols();
    std::array<uint16_t, 64> z{};
    z[0] = (uint16_t)(snn.prgs[2].int_buff[3] % P);
}

#ifdef SURFACE_NEGATIVE_CONTROL // must NOT compile: shows that the statements above bite (msb takes three arguments)
SAME(&ABY3Engine::msb, void (ABY3Engine::*)(const V2 &, B2 &));
#endif
