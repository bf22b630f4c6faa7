// Host-side helpers shared by the host translation units of libftk_hip.so (decoders, writers).
#pragma once

#include <functional>

namespace ftk_host {

// fn(0) runs on the caller, fn(1..n-1) on the library's persistent worker threads; returns when all are done.
void parallel_run(int n, const std::function<void(int)>& fn);
// The same on a SECOND pool of worker threads, for the consumer's side (results being widened or formatted while a
// decoder stream's regions - which are serialised on the first pool, one at a time, some 10 ms long - are running).
void parallel_run_results(int n, const std::function<void(int)>& fn);
// Threads a host-side parallel region uses by default: the cores this process may use (affinity and cgroup
// quota), at most 64.
int default_threads();
// Device blocks that outlive their user (the decoders' contig blocks, a context's scratch): take() hands out the
// smallest idle block of at least `bytes` on `device` or allocates one (nullptr: out of memory), give() keeps it for
// the next taker (ftk_cache_trim releases the idle ones).  The giver has synchronised whatever used the block.
void* device_block_take(size_t bytes, int device, size_t* cap_out);
void device_block_give(void* p, size_t cap, int device);
// Message behind ftk_fragtable_error() (thread-local), for host entry points that have no ctx.
void set_decode_error(const char* msg);

}  // namespace ftk_host
