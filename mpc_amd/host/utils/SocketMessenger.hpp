// TCP messenger, wire-compatible with the reference's utils/SocketMessenger.{hpp,cpp}: party i listens on
// 127.0.0.1:(PORT+i), connects to the two others, the connecting side first sends its 8-byte party id;
// one round sends, per peer, an 8-byte little-endian length followed by the payload, and receives the
// same framing from each peer.  Unlike a send-everything-then-receive loop, a round here moves all four
// byte streams concurrently (poll + non-blocking I/O), so payloads far larger than the kernel's socket
// buffers (the 128 MiB - 2 GiB matrix messages) cannot deadlock.
#pragma once
#include <stddef.h>

#include "VirtuleMessenger.hpp"

#define PORT 8000 // party i listens on PORT + i

class SocketMessenger : public VirtuleMessenger {
  public:
    explicit SocketMessenger(size_t _party_id); // blocks until both peers are connected
    SocketMessenger(const SocketMessenger &others) = delete;
    ~SocketMessenger() override;

    void send_and_recv(const vector<string> &message_send, vector<string> &message_recv) override;

    // the reference keeps these public
    size_t party_id;
    int server_fd;  // listening socket
    int recv_fd[3]; // accepted connections, by peer
    int send_fd[3]; // outgoing connections, by peer
};
