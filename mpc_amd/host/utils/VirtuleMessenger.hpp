// One synchronous MPC round among the three parties: same interface as the reference's
// utils/VirtuleMessenger.hpp:6-16, so existing messengers and engines interoperate.
#pragma once
#include <string>
#include <vector>
using std::string;
using std::vector;
class VirtuleMessenger {
  public:
    VirtuleMessenger() = default;
    /// message_send: 3 payloads (the slot of the caller itself is ignored); after the call
    /// message_recv holds the 3 payloads received (own slot empty).  Every party must call it the same
    /// number of times, in lockstep.
    virtual void send_and_recv(const vector<string> &message_send, vector<string> &message_recv) = 0;
    virtual ~VirtuleMessenger() = default;
};
