// Three party threads exchange sixty rounds of payloads of many sizes (empty to 3 MiB, the sizes at which the
// library's multi-threaded host copy splits a payload) through InProcMessenger and check every message they receive.
// Needs no GPU: the messenger only uses mpcx_host_copy.  (Also the program to build with -fsanitize=thread.)
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../utils/InProcMessenger.hpp"
int main() {
    InProcHub hub;
    int bad = 0;
    auto party = [&](size_t id) {
        InProcMessenger m(&hub, id);
        vector<string> snd(3), rcv(3);
        for (size_t r = 0; r < 60; r++) {
            for (size_t to = 0; to < 3; to++) {
                const size_t n = (r % 7 == 3) ? (3u << 20) + r : (r * 977 + to * 13 + id) % 5000;
                snd[to].assign(n, (char)(id * 16 + to + r));
            }
            m.send_and_recv(snd, rcv);
            for (size_t from = 0; from < 3; from++) {
                if (from == id) continue;
                const size_t n = (r % 7 == 3) ? (3u << 20) + r : (r * 977 + id * 13 + from) % 5000;
                if (rcv[from].size() != n) { __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED); continue; }
                for (size_t i = 0; i < n; i += 997) if (rcv[from][i] != (char)(from * 16 + id + r)) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            }
        }
    };
    std::thread a(party, 0), b(party, 1), c(party, 2);
    a.join(); b.join(); c.join();
    printf(bad ? "inproc_selftest: %d bad payloads\n" : "inproc_selftest: all checks passed\n", bad);
    return bad != 0;
}
