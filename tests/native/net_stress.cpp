// CPU-only stress of the library's multi-threaded HOST code, built with ThreadSanitizer and with AddressSanitizer +
// UndefinedBehaviorSanitizer by tests/test_sanitizers.py (GPU AddressSanitizer is not available on the pool: the sanitizers
// run on the CPU build only).
//
//   net_stress pool
//       zk::HostPool (csrc/hostpool.hpp): tasks submitted from several threads at once, tasks that fan out to the pool
//       under the idle() rule, futures joined out of order, destruction with work queued.
//   net_stress net <world> <rounds>
//       zk::Net (csrc/net.hpp) in host mode over the shared-memory transport: <world> rank PROCESSES (forked before any
//       thread exists), each driving the three MultiplexedStreamID channels from three THREADS at once -- what
//       ext_wit.rs:158-170 does with its three joined d_ifft / d_fft -- through enter / gather / scatter / alltoall /
//       gather_host / bcast_host (mpc-net/src/lib.rs:89-176), checking every byte that arrives.
// Exit code 0 = every check passed (a sanitizer report makes the process exit non-zero by itself).
#include <sys/wait.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hostpool.hpp"
#include "net.hpp"

static int pool_mode() {
  std::atomic<long> sum{0};
  {
    zk::HostPool pool(6);
    std::vector<std::thread> subs;
    std::vector<std::vector<std::future<void>>> futs(4);
    for (int t = 0; t < 4; t++)
      subs.emplace_back([&, t]() {
        for (int i = 0; i < 400; i++)
          futs[t].push_back(pool.submit([&sum, &pool, i]() {
            sum.fetch_add(i, std::memory_order_relaxed);
            // fan out only with free workers for every sub-task (the rule msm_fold follows), then wait for them
            if (i % 50 == 0 && pool.idle() >= 2) {
              auto a = pool.submit([&sum]() { sum.fetch_add(1000000, std::memory_order_relaxed); });
              auto b = pool.submit([&sum]() { sum.fetch_add(1000000, std::memory_order_relaxed); });
              a.wait();
              b.wait();
            }
          }));
      });
    for (auto& th : subs) th.join();
    for (int t = 3; t >= 0; t--)
      for (size_t i = futs[t].size(); i-- > 0;) futs[t][i].wait();
    // destruction with work still queued: the pool drains it
    for (int i = 0; i < 64; i++) (void)pool.submit([&sum]() { sum.fetch_add(7, std::memory_order_relaxed); });
  }
  const long base = 4L * (399 * 400 / 2) + 64 * 7;
  const long got = sum.load();
  if (got < base || (got - base) % 1000000 != 0) {
    fprintf(stderr, "pool: sum %ld (base %ld)\n", got, base);
    return 1;
  }
  return 0;
}

static uint64_t pat(int rank, int sid, int round, size_t i) {
  return 0x9E3779B97F4A7C15ull * (uint64_t)(rank + 1) + 0x100000001B3ull * (uint64_t)(sid + 1) + 1315423911ull * (uint64_t)round + i;
}

static int rank_main(int rank, int world, int rounds, const unsigned char* id) {
  zk::Net net;
  const int n = 8, k = n / world;
  int rc = net.open(ZK_NET_SHM, rank, world, n, -1, true, id, (size_t)4 << 20);
  if (rc) {
    fprintf(stderr, "rank %d: open failed: %s\n", rank, net.err.c_str());
    return 2;
  }
  net.timeout_ms = 20000;
  std::atomic<int> bad{0};
  auto chan = [&](int sid) {
    const size_t words = (size_t)(3000 + 1111 * sid);          // per party row; > one staging chunk on channel 2
    const size_t row_bytes = words * 8, mine_bytes = row_bytes * k;
    std::vector<uint64_t> mine(words * k), full(rank == 0 ? words * n : 0), back(words * k);
    std::vector<uint64_t> a2a_s(words * world), a2a_r(words * world);
    for (int r = 0; r < rounds; r++) {
      uint32_t mask = 0;
      if (net.enter(sid, &mask) || mask != (world >= 32 ? 0xffffffffu : (1u << world) - 1)) {
        bad++;
        return;
      }
      for (size_t i = 0; i < mine.size(); i++) mine[i] = pat(rank, sid, r, i);
      if (net.gather(sid, mask, mine.data(), mine_bytes, rank == 0 ? full.data() : nullptr)) bad++;
      if (rank == 0) {
        for (int q = 0; q < world; q++)
          for (size_t i = 0; i < words * k; i++)
            if (full[(size_t)q * words * k + i] != pat(q, sid, r, i)) {
              bad++;
              break;
            }
        for (auto& v : full) v = ~v;                            // the king's answer: every rank's rows, complemented
      }
      if (net.scatter(sid, mask, rank == 0 ? full.data() : nullptr, mine_bytes, back.data())) bad++;
      for (size_t i = 0; i < back.size(); i++)
        if (back[i] != ~pat(rank, sid, r, i)) {
          bad++;
          break;
        }
      if (world > 1 && (r & 1)) {                               // all-to-all: block q of rank p arrives as block p at rank q
        if (net.enter(sid, &mask)) bad++;
        for (int q = 0; q < world; q++)
          for (size_t i = 0; i < words; i++) a2a_s[(size_t)q * words + i] = pat(rank * 16 + q, sid, r, i);
        if (net.alltoall(sid, mask, a2a_s.data(), row_bytes, a2a_r.data())) bad++;
        for (int q = 0; q < world; q++)
          for (size_t i = 0; i < words; i++)
            if (a2a_r[(size_t)q * words + i] != pat(q * 16 + rank, sid, r, i)) {
              bad++;
              break;
            }
      }
      // the small host messages of d_msm
      if (net.enter(sid, &mask)) bad++;
      uint64_t small[4] = {pat(rank, sid, r, 1), pat(rank, sid, r, 2), 0, 0};
      std::vector<uint64_t> all(4 * (size_t)world);
      if (net.gather_host(sid, mask, small, sizeof small, all.data())) bad++;
      uint64_t verdict[2] = {0, 0};
      if (rank == 0)
        for (int q = 0; q < world; q++) {
          if (all[4 * (size_t)q] != pat(q, sid, r, 1)) bad++;
          verdict[0] += all[4 * (size_t)q + 1];
        }
      if (net.bcast_host(sid, mask, verdict, sizeof verdict)) bad++;
      uint64_t want = 0;
      for (int q = 0; q < world; q++) want += pat(q, sid, r, 2);
      if (verdict[0] != want) bad++;
    }
  };
  std::thread t0(chan, 0), t1(chan, 1), t2(chan, 2);
  t0.join();
  t1.join();
  t2.join();
  if (bad.load()) fprintf(stderr, "rank %d: %d failed checks (%s)\n", rank, bad.load(), net.err.c_str());
  net.close();
  return bad.load() ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "pool")) return pool_mode();
  if (argc >= 4 && !strcmp(argv[1], "net")) {
    const int world = atoi(argv[2]), rounds = atoi(argv[3]);
    if (world < 1 || 8 % world) return 2;
    unsigned char id[ZK_NET_ID_BYTES];
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 2;
    fclose(f);
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {
      pid_t p = fork();                                         // before any thread exists in this process
      if (p == 0) _exit(rank_main(r, world, rounds, id));
      kids.push_back(p);
    }
    int rc = rank_main(0, world, rounds, id);
    for (pid_t p : kids) {
      int st = 0;
      waitpid(p, &st, 0);
      if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = rc ? rc : 1;
    }
    return rc;
  }
  fprintf(stderr, "usage: net_stress pool | net <world> <rounds>\n");
  return 2;
}
