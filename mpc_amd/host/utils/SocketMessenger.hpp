// TCP messenger, wire-compatible with the reference's utils/SocketMessenger.{hpp,cpp}: party i listens on
// 127.0.0.1:(PORT+i), connects to the two others, the connecting side first sends its 8-byte party id;
// one round sends, per peer, an 8-byte little-endian length followed by the payload, and receives the
// same framing from each peer.  Unlike a send-everything-then-receive loop, a round here moves all four
// byte streams concurrently (poll + non-blocking I/O), so payloads far larger than the kernel's socket
// buffers (the 128 MiB - 2 GiB matrix messages) cannot deadlock.
#pragma once
#include "VirtuleMessenger.hpp"
#include <stddef.h>
#define PORT 8000

class SocketMessenger : public VirtuleMessenger {
  public:
    size_t party_id;
    int server_fd;
    int recv_fd[3];
    int send_fd[3];
    SocketMessenger(size_t _party_id);
    SocketMessenger(const SocketMessenger &others) = delete;
    virtual void send_and_recv(const vector<string> &message_send, vector<string> &message_recv) override;
    virtual ~SocketMessenger();
};
