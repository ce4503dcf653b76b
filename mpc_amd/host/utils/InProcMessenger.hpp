// In-process VirtuleMessenger: the three parties are three threads of one process sharing a Hub.
// One round = post the addresses of the outgoing payloads in the hub, barrier, every receiver copies its payloads out
// of the senders' buffers (one copy per message, by the library's multi-threaded host copy), barrier.
// (The reference ships no in-process messenger; its tests use three OS processes over TCP.)
#pragma once
#include <condition_variable>
#include <mutex>

#include "../../../include/mpcx.h"
#include "VirtuleMessenger.hpp"

class InProcHub {
  public:
    const string *box[3][3] = {}; // box[from][to]: the sender's own buffer, valid until the round's second barrier
    size_t bytes_sent[3] = {0, 0, 0};
    size_t rounds[3] = {0, 0, 0};
    void arrive() {
        std::unique_lock<std::mutex> lk(mu_);
        const size_t gen = generation_;
        if (++waiting_ == 3) {
            waiting_ = 0;
            generation_++;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return generation_ != gen; });
        }
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    size_t waiting_ = 0, generation_ = 0;
};

class InProcMessenger : public VirtuleMessenger {
  public:
    InProcHub *hub;
    size_t party_id;
    InProcMessenger(InProcHub *_hub, size_t _party_id) : hub(_hub), party_id(_party_id) {}
    void send_and_recv(const vector<string> &message_send, vector<string> &message_recv) override {
        for (size_t to = 0; to < 3; to++)
            if (to != party_id) {
                hub->box[party_id][to] = &message_send[to];
                hub->bytes_sent[party_id] += message_send[to].size();
            }
        hub->rounds[party_id]++;
        hub->arrive();
        message_recv.resize(3);
        for (size_t from = 0; from < 3; from++)
            if (from != party_id) {
                const string &src = *hub->box[from][party_id];
                message_recv[from].resize(src.size());
                if (!src.empty()) mpcx_host_copy(&message_recv[from][0], src.data(), src.size());
            }
        message_recv[party_id].clear();
        hub->arrive();
    }
};
