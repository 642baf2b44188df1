// Host-side wait on a device event that neither sleeps away 35 us per round trip nor spins for ever.
//
// The eigensolver asks the device ~50 small questions per solve (csrc/eig.hip); hipStreamSynchronize parks the thread on
// an interrupt and costs ~35 us of idle GPU each time, so the answers are polled for.  An unbounded poll, however, turns
// a hung kernel into an R thread that burns a core for ever: the poll is therefore
// bounded -- after `spin_us` microseconds the thread falls back to the blocking wait, which also is what reports a
// device error.  Generic over the two operations so that the logic is tested on the CPU (tests/c/host_wait_check.cc).
#pragma once
#include <chrono>

namespace flgp {

// query():  0 = done, 1 = not ready yet, anything else = error code (returned as is)
// block():  blocking wait, returns 0 or an error code
// Returns 0 when the event has completed, else the error code.  *spun_out (optional) tells whether the fallback ran.
template <class Query, class Block>
inline int bounded_wait(Query query, Block block, long spin_us, bool *spun_out = nullptr) {
  if (spun_out) *spun_out = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned it = 0;; ++it) {
    const int q = query();
    if (q != 1) return q;
    // look at the clock every 64 polls: a poll is ~1 us of driver call, the clock read is not free either
    if ((it & 63u) == 63u) {
      const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      if (us >= spin_us) break;
    }
  }
  if (spun_out) *spun_out = true;
  return block();
}

}  // namespace flgp
