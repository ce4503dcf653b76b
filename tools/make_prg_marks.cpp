// Writes mpc_amd/prg_marks.bin: sparse checkpoints of the generator streams whose keys are compile-time constants of the
// engines (mpc_engine/ABY3Engine.hpp:10-15: KEY01, KEY12, KEY20 and the parties' own 0xF0) and of the streams the
// benchmark configurations draw their inputs from (PRgenerator(20 / 21 / 22 / 24, 0)).  One state every 2^22 stream
// words lets the library walk a long stream on many host threads at once instead of on one (prvhash-core64 has no
// jump-ahead: utils/PRgenerator.cpp:3-19).  The library trusts nothing in the file: a segment walked from mark k must
// end in the state of mark k + 1, or the serial walk takes over.
//   g++ -O2 -std=c++17 -pthread tools/make_prg_marks.cpp -o /tmp/make_prg_marks && /tmp/make_prg_marks mpc_amd/prg_marks.bin
// File: {magic, stride_words, nkeys} then per key {seed, lcg, count} and count states {seed, lcg, hash}; state k is the
// one before stream word k * stride (stream word 0 = int_buff[0], i.e. after the constructor's five discarded outputs).
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
typedef uint64_t u64;
struct St {
    u64 seed, lcg, hash;
};
static inline void step(St &s) {
    s.seed *= s.lcg * 2 + 1;
    const u64 rs = s.seed >> 32 | s.seed << 32;
    s.hash += rs + 0xAAAAAAAAAAAAAAAAull;
    s.lcg += s.seed + 0x5555555555555555ull;
    s.seed ^= s.hash;
}
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const u64 stride = 1ull << 22;
    struct Key {
        u64 seed, words;
    };
    // the engine keys to 2^33 words (a 2^24-element privateCompare reads buffer words up to 2^32, the 2^28-element
    // ReLU chain 7 * 2^28 pops), the input streams as far as the benchmark configurations read them
    const Key keys[] = {{0xAB01, 1ull << 33}, {0xBC12, 1ull << 33}, {0xCA20, 1ull << 33}, {0xF0, 1ull << 33},
                        {20, 1ull << 30}, {21, 1ull << 29}, {22, 1ull << 31}, {24, 1ull << 22}};
    const int nk = sizeof keys / sizeof keys[0];
    std::vector<std::vector<St>> marks(nk);
    std::vector<std::thread> th;
    for (int k = 0; k < nk; k++)
        th.emplace_back([&, k] {
            St s{keys[k].seed, 0, 0};
            for (int i = 0; i < 5; i++) step(s);
            const u64 count = keys[k].words / stride + 1;
            for (u64 m = 0; m < count; m++) {
                marks[k].push_back(s);
                if (m + 1 < count)
                    for (u64 i = 0; i < stride; i++) step(s);
            }
        });
    for (auto &t : th) t.join();
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 2;
    const u64 hdr[3] = {0x314B524D5843504Dull /* "MPCXMRK1" */, stride, (u64)nk};
    fwrite(hdr, 8, 3, f);
    for (int k = 0; k < nk; k++) {
        const u64 kh[3] = {keys[k].seed, 0, (u64)marks[k].size()};
        fwrite(kh, 8, 3, f);
        fwrite(marks[k].data(), sizeof(St), marks[k].size(), f);
    }
    fclose(f);
    printf("wrote %s\n", argv[1]);
    return 0;
}
