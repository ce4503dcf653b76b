// Host-side PRgenerator with the reference's public surface (utils/PRgenerator.hpp:7-24): public seed /
// lcg / hash / int_buff / bool_buff, pop_int64(), pop_bool().  It additionally remembers which stream it
// is (the constructor arguments) and how many words have been popped, so that the GPU kernels can
// regenerate exactly the words an engine method consumes and the object can be advanced past them.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <vector>
using std::vector;
static const size_t BUFF_NUM = 160000;
struct mpcx_ctx;
class PRgenerator {
  public:
    uint64_t seed;
    uint64_t lcg;
    uint64_t hash;
    vector<int64_t> int_buff;
    vector<int8_t> bool_buff;
    PRgenerator() = delete;
    int64_t pop_int64();
    int8_t pop_bool();
    PRgenerator(uint64_t _init_seed, uint64_t _init_lcg);
    ~PRgenerator() = default;

    // --- additions used by the GPU engines
    uint64_t init_seed, init_lcg; // stream identity
    uint64_t pops;                // words popped since construction
    /// the engine consumed `n` pops on the device: move this generator to the state after them
    void advance(mpcx_ctx *ctx, uint64_t n);
};
