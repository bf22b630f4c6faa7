// Host-side helpers shared by the host translation units of libftk_hip.so (decoders, writers).
#pragma once

#include <functional>

namespace ftk_host {

// fn(0) runs on the caller, fn(1..n-1) on the library's persistent worker threads; returns when all are done.
void parallel_run(int n, const std::function<void(int)>& fn);
// Threads a host-side parallel region uses by default: the cores this process may use (affinity and cgroup
// quota), at most 64.
int default_threads();
// Message behind ftk_fragtable_error() (thread-local), for host entry points that have no ctx.
void set_decode_error(const char* msg);

}  // namespace ftk_host
