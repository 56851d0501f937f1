// sdrx/SafeBufferPoolQueue.h -- drop-in for RadioNode/src/libraries/common/SafeBufferPoolQueue.h:15-47: the mutexed
// pool of BBP_Block buffers the reference's AM.h pulls in (TEST_Modulator_Demodulator.cpp:14 constructs one).
// Host plumbing at the boundary; no DSP.
#ifndef RADIONODE_SAFEBUFFERBOOLQUEUE_H
#define RADIONODE_SAFEBUFFERBOOLQUEUE_H

#include <mutex>
#include <queue>

#include "common.h"

class SafeBufferPoolQueue {
  public:
    SafeBufferPoolQueue(size_t size = 0)
    {
        for (size_t i = 0; i < size; i++) q.push(new BBP_Block());
    }
    ~SafeBufferPoolQueue()
    {
        std::lock_guard<std::mutex> lk(m);
        for (; !q.empty(); q.pop()) delete q.front();
    }
    void push(BBP_Block *elem)
    {
        std::lock_guard<std::mutex> lk(m);
        if (elem) q.push(elem);
    }
    BBP_Block *next() // nullptr when the pool is empty
    {
        std::lock_guard<std::mutex> lk(m);
        if (q.empty()) return nullptr;
        BBP_Block *b = q.front();
        q.pop();
        return b;
    }
    size_t get_available_blocks() const { return q.size(); }

  private:
    std::queue<BBP_Block *> q;
    std::mutex m;
};

#endif
