// CPU-only stress of the library's multi-threaded HOST code, built with ThreadSanitizer and with AddressSanitizer +
// UndefinedBehaviorSanitizer by tests/test_sanitizers.py (GPU AddressSanitizer is not available on the pool: the sanitizers
// run on the CPU build only).
//
//   net_stress pool
//       zk::HostPool (csrc/hostpool.hpp): tasks submitted from several threads at once, tasks that fan out to the pool
//       under the idle() rule, futures joined out of order, destruction with work queued.
//   net_stress gates <workers> <batches>
//       the launch / gate / fold structure of the batched prover (csrc/engine_groth16.inc.hpp batch_begin, csrc/
//       msm_impl.hpp MsmGate) with the device taken out: per batch three "launch" tasks chained V -> S+H -> W by flags the
//       later one spins on (bounded: a gate that is never raised is an error, not a hang), a "caller" thread that spins
//       on W's flag before the U launch and then joins the batch, ~200 small host-term tasks, and an S+H task that fans
//       out sub-tasks and waits for them with HostPool::wait_helping.  Run with 1, 2 and 4 workers and three batches in
//       flight: no schedule may block (every wait is on a flag raised by a task dequeued earlier, or helped).
//   net_stress rccl <world> <rounds>
//       the RCCL branches of zk::Net (group construction in gather / scatter / alltoall, csrc/net.hpp) with ncclSend /
//       ncclRecv replaced by shared-memory mailboxes (Rccl::stub): the same three channel threads per rank and the same
//       byte checks as `net`, plus assertions on the CALL SEQUENCE every rank issued -- operations only inside groups, the
//       king's receives / sends in ascending rank order, a send paired with a receive per peer in the all-to-all, totals
//       per channel equal to what the rounds imply.  RCCL with more than one rank needs one GPU per rank and has never run
//       on hardware (the pool's boxes have one GPU): this is what pins that code path until it does.
//   net_stress net <world> <rounds>
//       zk::Net (csrc/net.hpp) in host mode over the shared-memory transport: <world> rank PROCESSES (forked before any
//       thread exists), each driving the three MultiplexedStreamID channels from three THREADS at once -- what
//       ext_wit.rs:158-170 does with its three joined d_ifft / d_fft -- through enter / gather / scatter / alltoall /
//       gather_host / bcast_host (mpc-net/src/lib.rs:89-176), checking every byte that arrives.
// Exit code 0 = every check passed (a sanitizer report makes the process exit non-zero by itself).
#include <sys/mman.h>
#include <sys/wait.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
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

// ---- the batched prover's host-side structure without a device (see the header comment)
static int gates_mode(int workers, int batches) {
  using clk = std::chrono::steady_clock;
  std::atomic<long> terms{0}, subs{0};
  std::atomic<int> bad{0};
  struct Batch {
    std::atomic<int> flag[4];
    std::vector<std::future<void>> fut;
  };
  auto spin_until = [&](std::atomic<int>& f) {
    const auto dl = clk::now() + std::chrono::seconds(60);
    for (unsigned i = 0; !f.load(std::memory_order_acquire); i++) {
      std::this_thread::yield();
      if ((i & 1023) == 1023 && clk::now() > dl) return false;
    }
    return true;
  };
  {
    zk::HostPool pool(workers);
    constexpr int INFLIGHT = 3;
    std::vector<std::unique_ptr<Batch>> slot(INFLIGHT);
    auto begin = [&](Batch& B) {
      for (auto& f : B.flag) f.store(0, std::memory_order_relaxed);
      for (int id = 0; id < 3; id++)                              // V, S+H, W: launch (gated), then "fold"
        B.fut.push_back(pool.submit([&, id, pb = &B]() {
          if (id > 0 && !spin_until(pb->flag[id - 1])) bad.fetch_add(1);
          pb->flag[id].store(1, std::memory_order_release);       // the accumulate kernel is enqueued, its event recorded
          std::this_thread::sleep_for(std::chrono::microseconds(200));      // the fold's wait for the device
          if (id == 1) {                                          // s*S, r*H per proof: fan out, help while waiting
            std::vector<std::future<void>> sub;
            for (int b = 1; b < 8; b++) sub.push_back(pool.submit([&]() { subs.fetch_add(1, std::memory_order_relaxed); }));
            subs.fetch_add(1, std::memory_order_relaxed);
            for (auto& f : sub)
              if (!pool.wait_helping(f, clk::now() + std::chrono::seconds(60))) bad.fetch_add(1);
          }
        }));
      for (int i = 0; i < 200; i++) B.fut.push_back(pool.submit([&]() { terms.fetch_add(1, std::memory_order_relaxed); }));
      if (!spin_until(B.flag[2])) bad.fetch_add(1);               // the caller's U launch behind W's flag
      B.flag[3].store(1, std::memory_order_release);
    };
    auto join = [&](Batch& B) {
      for (auto& f : B.fut)
        if (f.wait_for(std::chrono::seconds(60)) != std::future_status::ready) bad.fetch_add(1);
      B.fut.clear();
    };
    for (int i = 0; i < batches; i++) {
      std::unique_ptr<Batch>& B = slot[i % INFLIGHT];
      if (B) join(*B);
      B.reset(new Batch());
      begin(*B);
    }
    for (auto& B : slot)
      if (B) join(*B);
  }
  if (bad.load() || terms.load() != 200L * batches || subs.load() != 8L * batches) {
    fprintf(stderr, "gates: %d waits timed out, %ld host terms, %ld sub-tasks\n", bad.load(), terms.load(), subs.load());
    return 1;
  }
  return 0;
}

static uint64_t pat(int rank, int sid, int round, size_t i) {
  return 0x9E3779B97F4A7C15ull * (uint64_t)(rank + 1) + 0x100000001B3ull * (uint64_t)(sid + 1) + 1315423911ull * (uint64_t)round + i;
}

// ---- stand-ins for ncclSend / ncclRecv: one single-slot mailbox per (communicator, source, destination) in memory shared by
// the forked ranks.  Operations are queued between ncclGroupStart and ncclGroupEnd (per thread, as in NCCL) and carried
// out at the end of the group, sends first -- RCCL runs a group's operations concurrently; with one mailbox per directed
// pair no order of them can block.
namespace stub {
constexpr size_t CAP = (size_t)512 << 10;
struct Mail {
  std::atomic<uint64_t> posted, consumed;
  size_t len;
  unsigned char data[CAP];
};
struct Op {
  bool send;
  int peer, comm;
  size_t bytes;
  void* buf;
};
static Mail* mail = nullptr;                // [NET_NSID][world][world]
static int world = 0, rank = 0;
static std::atomic<int> next_comm{0};
static thread_local std::vector<Op> group;
static thread_local int depth = 0;
static std::mutex log_mu;
static std::vector<std::vector<std::string>> calls(zk::NET_NSID);      // per communicator: "S", "s<peer>:<bytes>", "r..", "E"
static std::atomic<int> violations{0};
static Mail& box(int comm, int src, int dst) { return mail[((size_t)comm * world + src) * world + dst]; }
static bool wait_for(const std::function<bool()>& ready) {
  const auto dl = std::chrono::steady_clock::now() + std::chrono::seconds(30);
  for (unsigned i = 0; !ready(); i++) {
    std::this_thread::yield();
    if ((i & 255) == 255 && std::chrono::steady_clock::now() > dl) return false;
  }
  return true;
}
static void note(int comm, const std::string& what) {
  std::lock_guard<std::mutex> lk(log_mu);
  calls[comm].push_back(what);
}
static int GroupStart() {
  depth++;
  return 0;
}
static int queue_op(bool send, void* buf, size_t count, int peer, void* comm_, hipStream_t) {
  const int comm = (int)(intptr_t)comm_ - 1;
  if (depth < 1 || comm < 0 || comm >= zk::NET_NSID || peer < 0 || peer >= world || peer == rank || count > CAP) {
    violations++;
    return 5;                                                         // ncclInvalidUsage
  }
  group.push_back({send, peer, comm, count, buf});
  return 0;
}
static int Send(const void* b, size_t count, int, int peer, void* comm, hipStream_t st) { return queue_op(true, (void*)b, count, peer, comm, st); }
static int Recv(void* b, size_t count, int, int peer, void* comm, hipStream_t st) { return queue_op(false, b, count, peer, comm, st); }
static int GroupEnd() {
  if (--depth) return depth < 0 ? 5 : 0;
  std::vector<Op> ops;
  ops.swap(group);
  if (ops.empty()) return 0;
  note(ops[0].comm, "S");
  for (const Op& o : ops) note(o.comm, std::string(o.send ? "s" : "r") + std::to_string(o.peer) + ":" + std::to_string(o.bytes));
  note(ops[0].comm, "E");
  int rc = 0;
  for (const Op& o : ops)
    if (o.send) {
      Mail& m = box(o.comm, rank, o.peer);
      if (!wait_for([&] { return m.consumed.load(std::memory_order_acquire) == m.posted.load(std::memory_order_relaxed); })) rc = 6;
      memcpy(m.data, o.buf, o.bytes);
      m.len = o.bytes;
      m.posted.fetch_add(1, std::memory_order_release);
    }
  for (const Op& o : ops)
    if (!o.send) {
      Mail& m = box(o.comm, o.peer, rank);
      if (!wait_for([&] { return m.posted.load(std::memory_order_acquire) > m.consumed.load(std::memory_order_relaxed); })) {
        rc = 6;
        continue;
      }
      if (m.len != o.bytes) violations++, rc = 5;                     // the two sides disagree on the byte count
      else memcpy(o.buf, m.data, o.bytes);
      m.consumed.fetch_add(1, std::memory_order_release);
    }
  return rc;
}
static int CommInitRank(void** comm, int w, zk::Rccl::UniqueId, int r) {
  if (w != world || r != rank) return 5;
  *comm = (void*)(intptr_t)(1 + next_comm.fetch_add(1));
  return 0;
}
static int CommDestroy(void*) { return 0; }
static const char* ErrorString(int rc) { return rc == 6 ? "stub: peer did not show up" : "stub: invalid usage"; }
static void install(int world_, int rank_) {
  world = world_;
  rank = rank_;
  zk::Rccl& R = zk::Rccl::inst();
  R.stub = true;
  R.CommInitRank = CommInitRank;
  R.CommDestroy = CommDestroy;
  R.CommAbort = CommDestroy;
  R.GroupStart = GroupStart;
  R.GroupEnd = GroupEnd;
  R.Send = Send;
  R.Recv = Recv;
  R.GetErrorString = ErrorString;
}
// the call sequence of one rank on the three channels against what `rounds` rounds of gather + scatter (+ an all-to-all in
// odd rounds) imply; rows = transfers per rank and verb (k party rows with a party map, else 1)
static int check_calls(int rounds, int rows) {
  int bad = violations.load();
  for (int sid = 0; sid < 3; sid++) {
    const std::vector<std::string>& c = calls[sid];
    long sends = 0, recvs = 0, groups = 0;
    bool in = false;
    int last_peer = -1;
    std::vector<std::string> g;
    for (const std::string& e : c) {
      if (e == "S") {
        if (in) bad++;
        in = true, last_peer = -1, g.clear(), groups++;
      } else if (e == "E") {
        if (!in) bad++;
        in = false;
        // a group is all sends, all receives (star verbs: ascending peers) or send / receive pairs per peer (all-to-all)
        bool all_s = true, all_r = true, pairs = g.size() % 2 == 0;
        for (size_t i = 0; i < g.size(); i++) {
          all_s = all_s && g[i][0] == 's';
          all_r = all_r && g[i][0] == 'r';
          if (pairs && (i & 1) && (g[i][0] != 'r' || g[i - 1][0] != 's' || g[i].substr(1) != g[i - 1].substr(1))) pairs = false;
        }
        if (!all_s && !all_r && !pairs) bad++;
        if ((all_s || all_r) && rank != 0)
          for (const std::string& x : g)
            if (atoi(x.c_str() + 1) != 0) bad++;                       // a client of the star talks to the king only
      } else {
        if (!in) bad++;
        const int peer = atoi(e.c_str() + 1);
        if (peer < last_peer) bad++;                                   // ascending rank order inside a group
        last_peer = peer;
        (e[0] == 's' ? sends : recvs)++;
        g.push_back(e);
      }
    }
    const long a2a = world > 1 ? rounds / 2 : 0;                       // odd rounds
    const long star = rank == 0 ? (long)rounds * (world - 1) * rows : (long)rounds * rows;
    const long want_s = star + a2a * (world - 1), want_g = world > 1 ? 2L * rounds + a2a : 0;
    if (sends != want_s || recvs != want_s || groups != want_g) {
      fprintf(stderr, "rank %d channel %d: %ld sends, %ld receives, %ld groups; expected %ld, %ld, %ld\n", rank, sid, sends, recvs, groups,
              want_s, want_s, want_g);
      bad++;
    }
  }
  for (int c = 0; c < zk::NET_NSID; c++)
    for (int q = 0; q < world; q++)
      if (box(c, rank, q).posted.load() != box(c, rank, q).consumed.load()) bad++;     // nothing left in flight
  return bad;
}
}  // namespace stub

static int rank_main(int rank, int world, int rounds, const unsigned char* id, int transport = ZK_NET_SHM, const int* pmap = nullptr) {
  zk::Net net;
  const int n = 8, k = n / world;
  if (transport == ZK_NET_RCCL) stub::install(world, rank);
  int rc = net.open(transport, rank, world, n, -1, true, id, (size_t)4 << 20, pmap);
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
  if (transport == ZK_NET_RCCL) {
    const int cb = stub::check_calls(rounds, pmap ? k : 1);
    if (cb) fprintf(stderr, "rank %d: %d call-sequence violations\n", rank, cb);
    bad += cb;
  }
  if (bad.load()) fprintf(stderr, "rank %d: %d failed checks (%s)\n", rank, bad.load(), net.err.c_str());
  net.close();
  return bad.load() ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "pool")) return pool_mode();
  if (argc >= 4 && !strcmp(argv[1], "gates")) return gates_mode(atoi(argv[2]), atoi(argv[3]));
  if (argc >= 4 && (!strcmp(argv[1], "net") || !strcmp(argv[1], "rccl"))) {
    const int world = atoi(argv[2]), rounds = atoi(argv[3]);
    if (world < 1 || 8 % world) return 2;
    const bool rccl = !strcmp(argv[1], "rccl");
    const int transport = rccl ? ZK_NET_RCCL : ZK_NET_SHM;
    if (rccl) {
      const size_t bytes = sizeof(stub::Mail) * zk::NET_NSID * world * world;
      void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);      // zero-filled, shared by the forks
      if (m == MAP_FAILED) return 2;
      stub::mail = (stub::Mail*)m;
    }
    unsigned char id[ZK_NET_ID_BYTES];
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 2;
    fclose(f);
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {
      pid_t p = fork();                                         // before any thread exists in this process
      if (p == 0) _exit(rank_main(r, world, rounds, id, transport));
      kids.push_back(p);
    }
    int rc = rank_main(0, world, rounds, id, transport);
    for (pid_t p : kids) {
      int st = 0;
      waitpid(p, &st, 0);
      if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = rc ? rc : 1;
    }
    return rc;
  }
  fprintf(stderr, "usage: net_stress pool | gates <workers> <batches> | net <world> <rounds> | rccl <world> <rounds>\n");
  return 2;
}
