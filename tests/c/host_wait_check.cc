// CPU check of flgp::bounded_wait (csrc/host_wait.h): completes while polling, falls back to the blocking wait after the
// spin budget, and passes errors through from both operations.
#include "../../flgp_amd/csrc/host_wait.h"
#include <cstdio>
#include <thread>
int main() {
  int fails = 0;
  {  // ready on the 10th poll: no fallback
    int n = 0; bool spun = true; int blocks = 0;
    int rc = flgp::bounded_wait([&] { return ++n >= 10 ? 0 : 1; }, [&] { ++blocks; return 0; }, 1000, &spun);
    if (rc != 0 || spun || blocks != 0 || n != 10) { printf("FAIL ready-while-polling rc=%d spun=%d blocks=%d n=%d\n", rc, spun, blocks, n); ++fails; }
  }
  {  // never ready: the poll must give up after ~2 ms and call the blocking wait exactly once
    bool spun = false; int blocks = 0; long polls = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = flgp::bounded_wait([&] { ++polls; return 1; }, [&] { ++blocks; return 0; }, 2000, &spun);
    const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0 || !spun || blocks != 1 || us < 2000 || us > 500000) { printf("FAIL fallback rc=%d spun=%d blocks=%d us=%ld polls=%ld\n", rc, spun, blocks, us, polls); ++fails; }
  }
  {  // an error from the query is returned at once
    bool spun = true; int blocks = 0;
    int rc = flgp::bounded_wait([&] { return 700; }, [&] { ++blocks; return 0; }, 1000, &spun);
    if (rc != 700 || spun || blocks != 0) { printf("FAIL query error rc=%d\n", rc); ++fails; }
  }
  {  // an error from the blocking wait is returned
    int rc = flgp::bounded_wait([&] { return 1; }, [&] { return 719; }, 100);
    if (rc != 719) { printf("FAIL block error rc=%d\n", rc); ++fails; }
  }
  {  // becomes ready from another thread while polling
    volatile int flag = 0; bool spun = true;
    std::thread th([&] { std::this_thread::sleep_for(std::chrono::microseconds(300)); flag = 1; });
    int rc = flgp::bounded_wait([&] { return flag ? 0 : 1; }, [&] { return 5; }, 200000, &spun);
    th.join();
    if (rc != 0 || spun) { printf("FAIL cross-thread rc=%d spun=%d\n", rc, spun); ++fails; }
  }
  if (!fails) printf("host_wait ok\n");
  return fails;
}
