// TEST INFRASTRUCTURE ONLY.  Thin C wrapper around the *real* reference PRgenerator
// (compiled from /root/reference/utils/PRgenerator.cpp where it lies; see oracle/Makefile).
// Only the PRG can be built this way: the engines need Eigen 3.4.0, absent from the image.
// The output (oracle/_ref/libref_prg.so) is used to validate oracle/mpc_oracle.cpp's PRG
// restatement on arbitrary seeds and lengths.
#include "utils/PRgenerator.hpp"
#include <cstdint>

extern "C" {
// int_buff[first .. first+count) and bool_buff[...] of PRgenerator(seed, lcg), first+count <= BUFF_NUM
int ref_prg_buff(uint64_t seed, uint64_t lcg, uint64_t first, uint64_t count, int64_t *ints, int8_t *bools) {
    if (first + count > BUFF_NUM) return 1;
    PRgenerator g(seed, lcg);
    for (uint64_t i = 0; i < count; i++) {
        if (ints) ints[i] = g.int_buff[first + i];
        if (bools) bools[i] = g.bool_buff[first + i];
    }
    return 0;
}
// skip `first` pops, then `count` pops of pop_int64()
int ref_prg_pop(uint64_t seed, uint64_t lcg, uint64_t first, uint64_t count, int64_t *out) {
    PRgenerator g(seed, lcg);
    for (uint64_t i = 0; i < first; i++) g.pop_int64();
    for (uint64_t i = 0; i < count; i++) out[i] = g.pop_int64();
    return 0;
}
// alternating pop_int64 / pop_bool, as callers of the reference interleave them
int ref_prg_pop_mixed(uint64_t seed, uint64_t lcg, uint64_t count, int64_t *ints, int8_t *bools) {
    PRgenerator g(seed, lcg);
    for (uint64_t i = 0; i < count; i++) {
        ints[i] = g.pop_int64();
        bools[i] = g.pop_bool();
    }
    return 0;
}
// members after construction + `pops` pops
int ref_prg_state(uint64_t seed, uint64_t lcg, uint64_t pops, uint64_t *st) {
    PRgenerator g(seed, lcg);
    for (uint64_t i = 0; i < pops; i++) g.pop_int64();
    st[0] = g.seed;
    st[1] = g.lcg;
    st[2] = g.hash;
    return 0;
}
}
