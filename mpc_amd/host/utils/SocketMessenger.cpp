#include "SocketMessenger.hpp"

#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <thread>
#include <unistd.h>

namespace {
[[noreturn]] void die(const char *what) {
    perror(what);
    exit(EXIT_FAILURE);
}
void write_all(int fd, const void *buf, size_t len) {
    const char *p = (const char *)buf;
    while (len) {
        ssize_t k = send(fd, p, len, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            die("send");
        }
        p += k;
        len -= (size_t)k;
    }
}
void read_all(int fd, void *buf, size_t len) {
    char *p = (char *)buf;
    while (len) {
        ssize_t k = recv(fd, p, len, 0);
        if (k < 0) {
            if (errno == EINTR) continue;
            die("recv");
        }
        if (k == 0) {
            fprintf(stderr, "peer closed the connection\n");
            exit(EXIT_FAILURE);
        }
        p += k;
        len -= (size_t)k;
    }
}
// one direction of one peer link during a round: an 8-byte length, then the payload
struct Lane {
    int fd = -1;
    uint64_t header = 0;
    size_t header_done = 0;
    char *body = nullptr;
    size_t body_len = 0, body_done = 0;
    bool finished() const { return header_done == 8 && body_done == body_len; }
};
} // namespace

SocketMessenger::SocketMessenger(size_t _party_id) : party_id{_party_id} {
    for (size_t i = 0; i < 3; i++) recv_fd[i] = send_fd[i] = -1;
    int opt = 1;
    struct sockaddr_in address;
    memset(&address, 0, sizeof address);
    address.sin_family = AF_INET;
    address.sin_addr.s_addr = INADDR_ANY;
    address.sin_port = htons(PORT + party_id);
    if ((server_fd = socket(AF_INET, SOCK_STREAM, 0)) < 0) die("socket failed");
    if (setsockopt(server_fd, SOL_SOCKET, SO_REUSEADDR | SO_REUSEPORT, &opt, sizeof(opt))) die("setsockopt");
    if (bind(server_fd, (struct sockaddr *)&address, sizeof(address)) < 0) die("bind failed");
    if (listen(server_fd, 2) < 0) die("listen");

    std::thread dialer([this] {
        for (size_t i = 0; i < 3; i++) {
            if (i == party_id) continue;
            struct sockaddr_in peer;
            memset(&peer, 0, sizeof peer);
            peer.sin_family = AF_INET;
            peer.sin_port = htons(PORT + i);
            if (inet_pton(AF_INET, "127.0.0.1", &peer.sin_addr) <= 0) die("inet_pton");
            for (;;) { // the peer may not be listening yet
                int fd = socket(AF_INET, SOCK_STREAM, 0);
                if (fd < 0) die("socket");
                if (connect(fd, (struct sockaddr *)&peer, sizeof(peer)) == 0) {
                    send_fd[i] = fd;
                    break;
                }
                close(fd);
                usleep(2000);
            }
            int one = 1;
            setsockopt(send_fd[i], IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            uint64_t me = party_id;
            write_all(send_fd[i], &me, sizeof me);
        }
    });
    for (size_t k = 0; k < 2; k++) {
        int fd = accept(server_fd, NULL, NULL);
        if (fd < 0) die("accept");
        uint64_t remote = 0;
        read_all(fd, &remote, sizeof remote);
        if (remote > 2 || remote == party_id || recv_fd[remote] != -1) {
            fprintf(stderr, "wrong remote_id\n");
            exit(EXIT_FAILURE);
        }
        recv_fd[remote] = fd;
    }
    dialer.join();
}

SocketMessenger::~SocketMessenger() {
    for (size_t i = 0; i < 3; i++) {
        if (i == party_id) continue;
        if (send_fd[i] >= 0) close(send_fd[i]);
        if (recv_fd[i] >= 0) close(recv_fd[i]);
    }
    shutdown(server_fd, SHUT_RDWR);
    close(server_fd);
}

void SocketMessenger::send_and_recv(const vector<string> &message_send, vector<string> &message_recv) {
    message_recv.resize(3);
    Lane out[2], in[2];
    size_t peers[2], np = 0;
    for (size_t i = 0; i < 3; i++)
        if (i != party_id) peers[np++] = i;
    for (size_t k = 0; k < 2; k++) {
        const size_t p = peers[k];
        out[k].fd = send_fd[p];
        out[k].header = message_send[p].size();
        out[k].body = (char *)message_send[p].data();
        out[k].body_len = message_send[p].size();
        in[k].fd = recv_fd[p];
    }
    message_recv[party_id].clear();
    for (;;) {
        struct pollfd pf[4];
        int n = 0, idx[4];
        for (int k = 0; k < 2; k++) {
            if (!out[k].finished()) {
                pf[n] = {out[k].fd, POLLOUT, 0};
                idx[n++] = k;
            }
            if (!in[k].finished() || in[k].header_done < 8) {
                pf[n] = {in[k].fd, POLLIN, 0};
                idx[n++] = 2 + k;
            }
        }
        if (n == 0) break;
        if (poll(pf, n, -1) < 0) {
            if (errno == EINTR) continue;
            die("poll");
        }
        for (int q = 0; q < n; q++) {
            if (!(pf[q].revents & (POLLIN | POLLOUT | POLLHUP | POLLERR))) continue;
            if (idx[q] < 2) {
                Lane &L = out[idx[q]];
                const char *src = L.header_done < 8 ? (const char *)&L.header + L.header_done : L.body + L.body_done;
                size_t len = L.header_done < 8 ? 8 - L.header_done : L.body_len - L.body_done;
                ssize_t k = send(L.fd, src, len, MSG_DONTWAIT | MSG_NOSIGNAL);
                if (k < 0) {
                    if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) continue;
                    die("send");
                }
                if (L.header_done < 8) L.header_done += (size_t)k;
                else L.body_done += (size_t)k;
            } else {
                Lane &L = in[idx[q] - 2];
                char *dst = L.header_done < 8 ? (char *)&L.header + L.header_done : L.body + L.body_done;
                size_t len = L.header_done < 8 ? 8 - L.header_done : L.body_len - L.body_done;
                ssize_t k = recv(L.fd, dst, len, MSG_DONTWAIT);
                if (k < 0) {
                    if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) continue;
                    die("recv");
                }
                if (k == 0) {
                    fprintf(stderr, "peer closed the connection\n");
                    exit(EXIT_FAILURE);
                }
                if (L.header_done < 8) {
                    L.header_done += (size_t)k;
                    if (L.header_done == 8) { // length known: size the destination string
                        string &dstr = message_recv[peers[idx[q] - 2]];
                        dstr.resize(L.header);
                        L.body = &dstr[0];
                        L.body_len = L.header;
                    }
                } else {
                    L.body_done += (size_t)k;
                }
            }
        }
    }
}
