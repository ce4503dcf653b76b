// SocketMessenger self-test (CPU only): three parties as three threads on loopback exchange payloads
// far larger than the kernel socket buffers in one round (a blocking send-all-then-receive-all round
// deadlocks there), plus empty payloads, and verify every byte.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>

#include "../utils/SocketMessenger.hpp"

static std::atomic<int> bad{0};

static char pattern(size_t from, size_t to, size_t round, size_t i) { return (char)(i * 31 + from * 7 + to * 3 + round); }

static void party(size_t id, size_t big) {
    SocketMessenger m(id);
    vector<string> send(3), recv(3);
    const size_t sizes[4] = {0, 17, big, 1};
    for (size_t round = 0; round < 4; round++) {
        for (size_t to = 0; to < 3; to++) {
            if (to == id) continue;
            const size_t len = sizes[(round + id + to) % 4];
            send[to].resize(len);
            for (size_t i = 0; i < len; i++) send[to][i] = pattern(id, to, round, i);
        }
        m.send_and_recv(send, recv);
        for (size_t from = 0; from < 3; from++) {
            if (from == id) continue;
            const size_t len = sizes[(round + from + id) % 4];
            if (recv[from].size() != len) {
                bad++;
                continue;
            }
            for (size_t i = 0; i < len; i++)
                if (recv[from][i] != pattern(from, id, round, i)) {
                    bad++;
                    break;
                }
        }
    }
}

int main(int argc, char **argv) {
    const size_t big = argc > 1 ? strtoull(argv[1], NULL, 10) : (64u << 20);
    std::thread t0(party, 0, big), t1(party, 1, big), t2(party, 2, big);
    t0.join();
    t1.join();
    t2.join();
    printf("socket_selftest: %s (%zu-byte payloads)\n", bad.load() ? "FAILED" : "ok", big);
    return bad.load() ? 1 : 0;
}
