// Host-side generator object with the public surface callers of the reference use (utils/PRgenerator.hpp:7-24):
// the three state words, the two constructor-filled buffers, pop_int64() and pop_bool().  It also remembers
// which stream it is (the constructor arguments) and how many words have been popped, so that the GPU kernels
// can regenerate exactly the words an engine method consumes and the object can be moved past them.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <vector>

using std::vector;

static const size_t BUFF_NUM = 160000; // entries of each buffer

struct mpcx_ctx;

class PRgenerator {
  public:
    PRgenerator() = delete;
    PRgenerator(uint64_t _init_seed, uint64_t _init_lcg); // 5 discarded outputs, then the two buffers are filled
    ~PRgenerator() = default;

    int64_t pop_int64(); // the next output
    int8_t pop_bool();   // the low bit of the next output

    // generator state after the last pop
    uint64_t seed, lcg, hash;
    // int_buff[i] / bool_buff[i]: outputs 2i and 2i+1 (low bit) after the discarded ones
    vector<int64_t> int_buff;
    vector<int8_t> bool_buff;

    // ---- additions for the GPU engines
    uint64_t init_seed, init_lcg; // which stream this is
    uint64_t pops;                // words popped since construction
    // the engine consumed n pops on the device: put this generator into the state after them
    void advance(mpcx_ctx *ctx, uint64_t n);
};
