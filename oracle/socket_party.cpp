// Test infrastructure: one party of a three-party exchange over a SocketMessenger, as its own process.
// The messenger class comes from the header named by SOCKET_MESSENGER_HEADER, so the same program is built
// twice: against the reference's utils/SocketMessenger.{hpp,cpp} where they lie (-> oracle/_ref/ref_socket_party)
// and against this repository's mpc_amd/host/utils/SocketMessenger (-> oracle/my_socket_party).  Parties of the two
// builds must be able to talk to each other in any mix: same ports, same handshake, same framing
// (utils/SocketMessenger.cpp:34-164).
//
//   socket_party <party_id> [rounds]     exit status 0 = every received byte was the expected one
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
using std::string;
using std::vector;
#include SOCKET_MESSENGER_HEADER

// payload of `from` for `to` in `round`: length and bytes are functions of the triple alone; kept far below the
// kernel's socket buffers because the reference sends to both peers before it receives anything
static size_t length(size_t from, size_t to, size_t round) { return ((from * 7 + to * 3 + round * 5) % 4 == 0) ? 0 : 1 + (from * 1013 + to * 337 + round * 2741) % 20000; }
static char byte_at(size_t from, size_t to, size_t round, size_t i) { return (char)(i * 31 + from * 7 + to * 3 + round * 11); }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const size_t id = strtoul(argv[1], NULL, 10), rounds = argc > 2 ? strtoul(argv[2], NULL, 10) : 6;
    if (id > 2) return 2;
    SocketMessenger m(id);
    vector<string> send(3), recv(3);
    int bad = 0;
    for (size_t round = 0; round < rounds; round++) {
        for (size_t to = 0; to < 3; to++) {
            send[to].clear();
            if (to == id) continue;
            send[to].resize(length(id, to, round));
            for (size_t i = 0; i < send[to].size(); i++) send[to][i] = byte_at(id, to, round, i);
        }
        m.send_and_recv(send, recv);
        for (size_t from = 0; from < 3; from++) {
            if (from == id) continue;
            if (recv[from].size() != length(from, id, round)) {
                bad++;
                continue;
            }
            for (size_t i = 0; i < recv[from].size(); i++)
                if (recv[from][i] != byte_at(from, id, round, i)) {
                    bad++;
                    break;
                }
        }
    }
    printf("party %zu: %zu rounds, %d bad payloads\n", id, rounds, bad);
    return bad ? 1 : 0;
}
