#include "PRgenerator.hpp"

#include <stdio.h>

#include "../../../include/mpcx.h"

namespace {
// one prvhash-core64 round (the recurrence the reference uses, utils/PRgenerator.cpp:3-19)
inline uint64_t next_word(uint64_t &seed, uint64_t &lcg, uint64_t &hash) {
    seed *= lcg * 2 + 1;
    const uint64_t rot = (seed >> 32) | (seed << 32);
    hash += rot + 0xAAAAAAAAAAAAAAAAull;
    lcg += seed + 0x5555555555555555ull;
    seed ^= hash;
    return lcg ^ rot;
}
} // namespace

PRgenerator::PRgenerator(uint64_t _init_seed, uint64_t _init_lcg)
    : seed(_init_seed), lcg(_init_lcg), hash(0), int_buff(BUFF_NUM), bool_buff(BUFF_NUM), init_seed(_init_seed),
      init_lcg(_init_lcg), pops(0) {
    for (int warm = 0; warm < 5; warm++) next_word(seed, lcg, hash);
    for (size_t i = 0; i < BUFF_NUM; i++) {
        int_buff[i] = (int64_t)next_word(seed, lcg, hash);
        bool_buff[i] = (int8_t)(next_word(seed, lcg, hash) & 1);
    }
}

int64_t PRgenerator::pop_int64() {
    pops++;
    return (int64_t)next_word(seed, lcg, hash);
}

int8_t PRgenerator::pop_bool() {
    pops++;
    return (int8_t)(next_word(seed, lcg, hash) & 1);
}

void PRgenerator::advance(mpcx_ctx *ctx, uint64_t n) {
    if (n == 0) return;
    uint64_t st[3];
    if (mpcx_prg_state_at_pop(ctx, init_seed, init_lcg, pops + n, st)) {
        fprintf(stderr, "PRgenerator::advance: %s\n", mpcx_last_error(ctx));
        abort();
    }
    pops += n;
    seed = st[0];
    lcg = st[1];
    hash = st[2];
}
