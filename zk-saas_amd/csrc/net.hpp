// The MPC star network of the reference on one multi-GPU node, inside the library.
//
// Reference: mpc-net/src/lib.rs:43-53 (`MpcNet`: n_parties, party_id, is_king, per-channel `MultiplexedStreamID`),
// :89-135 `client_send_or_king_receive` (every party sends to the king; the king waits up to 30 s per peer and
// otherwise reports the peer as timed out), :137-176 `client_receive_or_king_send` (the king sends each party ITS
// OWN answer, equal lengths enforced), and mpc-net/src/ser_net.rs:16-120 (the serialising wrappers; `Partial`
// results when some parties are missing, ser_net.rs:57-94).  One process drives one GPU and the k = n / world
// parties mapped to it: by default the contiguous blocks (party p on rank p / k), or any balanced `party_to_rank` map
// (MpcNet ids are arbitrary, lib.rs:43-53).  The king's work is done by rank 0, whichever parties it holds.  The king
// always sees the rows of the parties present in ASCENDING PARTY ORDER and answers with one row per party id, so the
// king kernels do not know the map: with a non-contiguous map gather / scatter move party rows one by one.
//
// Two planes:
//   control  a small POSIX shared-memory block (all ranks are on one node): per channel an arrival word per rank, the
//            king's verdict (which ranks take part in this round -- the emulation of mpc-net's timeout / `Partial`
//            semantics: a rank that does not enter a collective within the timeout is left out, the king continues
//            through lagrange_unpack if enough parties remain, the late rank gets ZK_ERR_PROTOCOL), and a small
//            payload area for host-side values (the partial points of d_msm);
//   data     transport RCCL: ncclSend / ncclRecv in one group per round on the channel's own communicator and HIP
//            stream (xGMI, GPU-resident payload, raw Montgomery limbs -- nothing is serialised), only among the ranks
//            of the verdict, so a missing rank cannot hang the others;
//            transport SHM: the same verbs staged through a shared-memory segment (D2H, H2D); used by the tests (two
//            ranks on one GPU, or no GPU at all: with a NULL context the buffers are host memory) and as a
//            fallback where RCCL cannot run.
//            transport LOCAL: world = 1, plain device copies.
// Three channels can be in flight at once (ext_wit.rs:158-170 joins three d_ifft / d_fft on CHANNEL0-2): each has its
// own communicator, stream, sequence counter and staging segment.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zksaas.h"

namespace zk {

constexpr int NET_MAXR = 16;          // ranks (<= n parties)
constexpr int NET_NSID = 4;           // channels: 0..2 = MultiplexedStreamID::{Zero,One,Two}, 3 = internal
constexpr int NET_PAYLOAD = 4096;     // host payload bytes per rank and channel
constexpr uint32_t NET_MAGIC = 0x7a6b6e31;

struct NetChan {
  std::atomic<uint64_t> arrive[NET_MAXR];
  std::atomic<uint64_t> verdict_seq;
  std::atomic<uint32_t> verdict_mask;
  std::atomic<uint64_t> rank_tick[NET_MAXR];     // data plane SHM: rank's slot written (gather) / consumed (scatter)
  std::atomic<uint64_t> king_tick;               // king consumed the gather slots / filled the scatter slots
  std::atomic<uint64_t> pay_tick[NET_MAXR];
  std::atomic<uint64_t> pay_king;
  std::atomic<uint64_t> a2a_post[NET_MAXR];      // all-to-all SHM: rank's outgoing blocks written / every block for it read
  std::atomic<uint64_t> a2a_done[NET_MAXR];
  unsigned char payload[NET_MAXR][NET_PAYLOAD];
  unsigned char pay_out[NET_PAYLOAD];
};
struct NetCtl {
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> attached;
  NetChan chan[NET_NSID];
};

// librccl entry points, resolved at run time (the PyTorch wheel ships its own copy: whichever copy is already
// loaded in the process is used, so there is one RCCL per process)
struct Rccl {
  typedef int (*get_uid_t)(void*);
  typedef int (*init_rank_t)(void**, int, struct UniqueId, int);
  struct UniqueId {
    char internal[128];
  };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  // TEST HARNESS ONLY (tests/native/net_stress.cpp): the table was filled with stand-ins for ncclSend / ncclRecv so that the
  // RCCL branches of gather / scatter / alltoall -- group construction, peers, byte counts -- run on a box without GPUs
  // (host mode).  Nothing in the library sets it.
  bool stub = false;
  bool load(std::string* err) {
    if (handle || stub) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
      if (handle) break;
    }
    // Not yet in the process: load it PRIVATELY.  A PyTorch-ROCm wheel bundles its own librccl + librocm_smi64, and
    // `import torch` AFTER this point brings those in beside /opt/rocm's: two copies of rocm_smi with the same global
    // objects, which (bound to one another through the global scope) are destroyed twice at exit -- "double free or
    // corruption", status 134, seen at the end of the whole GPU test suite in round 6 (test_gpu_dist's RCCL net first,
    // test_gpu_hardening's `import torch` later; backtrace: ~map in librocm_smi64's exit handlers).  RTLD_LOCAL keeps this
    // copy's symbols out of the global scope (a later copy binds to itself), RTLD_DEEPBIND makes this copy and its
    // dependencies prefer their own definitions over any copy already there.  The HIP runtime is shared either way (same
    // SONAME: the loader reuses the one in the process).
    if (!handle)
      for (const char* nm : names) {
        handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (handle) break;
      }
    if (!handle) {
      *err = std::string("librccl not found: ") + dlerror();
      return false;
    }
    auto sym = [&](const char* s) { return dlsym(handle, s); };
    GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
    CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    CommAbort = (int (*)(void*))sym("ncclCommAbort");
    GroupStart = (int (*)())sym("ncclGroupStart");
    GroupEnd = (int (*)())sym("ncclGroupEnd");
    Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
    Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
    GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv) {
      *err = "librccl lacks a required symbol";
      return false;
    }
    return true;
  }
  static Rccl& inst() {
    static Rccl r;
    return r;
  }
};

inline uint64_t net_now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000 + (uint64_t)ts.tv_nsec / 1000000;
}

class Net {
 public:
  int transport = ZK_NET_LOCAL, rank = 0, world = 1, n = 0, device = -1;
  bool host_mode = false;                 // no GPU context: buffers are host memory (tests of the protocol flow)
  uint64_t timeout_ms = 30000;            // mpc-net/src/lib.rs:98 `timeout(Duration::from_secs(30), ..)`
  std::string err;
  int err_party = -1;

  ~Net() { close(); }

  int parties_per_rank() const { return n / world; }
  // party held in row i of rank r's buffers (rows of a rank are in ascending party order)
  int party(int r, int i) const { return contig_ ? r * parties_per_rank() + i : slot_party_[(size_t)r * parties_per_rank() + i]; }
  int first_party(int r) const { return party(r, 0); }
  bool contiguous() const { return contig_; }

  // id: NET_NSID RCCL unique ids (RCCL transport) or any 128-byte tag shared by the ranks (first 16 bytes name the
  // shared-memory block)
  int open(int transport_, int rank_, int world_, int n_, int device_, bool host_mode_, const unsigned char* id,
           size_t shm_bytes_per_chan, const int* party_to_rank = nullptr) {
    transport = transport_;
    rank = rank_;
    world = world_;
    n = n_;
    device = device_;
    host_mode = host_mode_;
    if (world < 1 || world > NET_MAXR || rank < 0 || rank >= world || n % world) return fail("bad rank / world size (world must divide n)");
    alive_.store(full_mask(), std::memory_order_relaxed);      // every rank counts as alive until a verdict leaves it out
    if (party_to_rank) {
      const int k = n / world;
      if (n > NET_MAXR * 64) return fail("too many parties for a party_to_rank map", ZK_ERR_BAD_INPUT);
      std::vector<int> cnt((size_t)world, 0);
      slot_party_.assign((size_t)n, -1);
      contig_ = true;
      for (int p = 0; p < n; p++) {
        const int r = party_to_rank[p];
        if (r < 0 || r >= world || cnt[r] >= k) return fail("party_to_rank must map n / world parties to every rank", ZK_ERR_BAD_INPUT);
        slot_party_[(size_t)r * k + cnt[r]++] = p;
        if (r != p / k) contig_ = false;
      }
      if (contig_) slot_party_.clear();
    }
    if (world == 1 && transport != ZK_NET_RCCL) transport = ZK_NET_LOCAL;
    if (transport == ZK_NET_LOCAL) {
      if (world != 1) return fail("the local transport needs world = 1");
      return make_streams();
    }
    if (!id) return fail("net id missing");
    if (transport != ZK_NET_RCCL && transport != ZK_NET_SHM) return fail("unknown transport");
    if (host_mode && transport == ZK_NET_RCCL && !Rccl::inst().stub) return fail("RCCL needs a GPU context");
    uint64_t hsh = 1469598103934665603ull;          // FNV-1a over the id: the name of the shared control block
    for (int i = 0; i < 128 * NET_NSID; i++) hsh = (hsh ^ id[i]) * 1099511628211ull;
    char tag[40];
    snprintf(tag, sizeof tag, "/zksaas_%016llx", (unsigned long long)hsh);
    name_ = tag;
    if (!map_ctl()) return ZK_ERR_NOT_CONNECTED;
    if (transport == ZK_NET_SHM) {
      cap_ = shm_bytes_per_chan ? shm_bytes_per_chan : ((size_t)64 << 20);
      cap_ = (cap_ / world) & ~(size_t)255;       // per rank slot
      for (int s = 0; s < NET_NSID; s++)
        if (!map_data(s, data_, "_d") || !map_data(s, adata_, "_a")) return ZK_ERR_NOT_CONNECTED;
    } else {
      Rccl& R = Rccl::inst();
      if (!R.load(&err)) return ZK_ERR_NOT_CONNECTED;
      for (int s = 0; s < NET_NSID; s++) {
        Rccl::UniqueId uid;
        memcpy(&uid, id + (size_t)s * 128, 128);
        int rc = R.CommInitRank(&comm_[s], world, uid, rank);
        if (rc) return fail(std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(rc) : "?"));
      }
    }
    if (int rc = make_streams()) return rc;
    // attach barrier: nobody proceeds before every rank mapped the block
    ctl_->attached.fetch_add(1);
    uint64_t dl = net_now_ms() + timeout_ms;
    while (ctl_->attached.load() < (uint32_t)world) {
      if (net_now_ms() > dl) return fail("timed out waiting for the other ranks to attach", ZK_ERR_NOT_CONNECTED);
      relax();
    }
    return ZK_OK;
  }

  int make_streams() {
    if (host_mode) return ZK_OK;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // king rounds are latency chains: highest priority
    for (int s = 0; s < NET_NSID; s++) {
      if (hipStreamCreateWithPriority(&stream_[s], hipStreamNonBlocking, hi) != hipSuccess) return fail("hipStreamCreate");
      if (hipEventCreateWithFlags(&ev_in_[s], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");
      if (hipEventCreateWithFlags(&ev_out_[s], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");
    }
    return ZK_OK;
  }

  void close() {
    if (transport == ZK_NET_RCCL)
      for (int s = 0; s < NET_NSID; s++)
        if (comm_[s]) {
          Rccl& R = Rccl::inst();
          if (aborted_ && R.CommAbort) (void)R.CommAbort(comm_[s]);
          else (void)R.CommDestroy(comm_[s]);
          comm_[s] = nullptr;
        }
    for (int s = 0; s < NET_NSID; s++) {
      if (stream_[s]) (void)hipStreamDestroy(stream_[s]);
      if (ev_in_[s]) (void)hipEventDestroy(ev_in_[s]);
      if (ev_out_[s]) (void)hipEventDestroy(ev_out_[s]);
      stream_[s] = nullptr;
      ev_in_[s] = ev_out_[s] = nullptr;
      if (data_[s]) munmap(data_[s], cap_ * world);
      if (adata_[s]) munmap(adata_[s], cap_ * world);
      data_[s] = adata_[s] = nullptr;
    }
    if (ctl_) munmap(ctl_, sizeof(NetCtl));
    ctl_ = nullptr;
    if (rank == 0 && !name_.empty()) {
      shm_unlink(name_.c_str());
      for (int s = 0; s < NET_NSID; s++) {
        shm_unlink((name_ + "_d" + std::to_string(s)).c_str());
        shm_unlink((name_ + "_a" + std::to_string(s)).c_str());
      }
    }
    name_.clear();
  }

  hipStream_t stream(int sid) const { return stream_[sid]; }
  // channel stream ordered after the caller's stream / caller's stream ordered after the channel stream
  int begin(int sid, hipStream_t caller) {
    if (host_mode) return ZK_OK;
    if (hipEventRecord(ev_in_[sid], caller) != hipSuccess || hipStreamWaitEvent(stream_[sid], ev_in_[sid], 0) != hipSuccess)
      return fail("stream ordering");
    return ZK_OK;
  }
  int end(int sid, hipStream_t caller) {
    if (host_mode) return ZK_OK;
    if (hipEventRecord(ev_out_[sid], stream_[sid]) != hipSuccess || hipStreamWaitEvent(caller, ev_out_[sid], 0) != hipSuccess)
      return fail("stream ordering");
    return ZK_OK;
  }

  // ---- control plane: enter a collective round on channel `sid`; returns the mask of participating ranks.
  // A rank that is not in the verdict gets ZK_ERR_PROTOCOL (it arrived after the king's timeout).
  // A rank the king once left out stays out: the king stops waiting for it (alive_), and the rank itself fails every
  // later round at once (dead_) instead of running one round behind the others on the channels it never entered.
  int enter(int sid, uint32_t* mask) {
    if (sid < 0 || sid >= NET_NSID) return fail("bad channel id", ZK_ERR_BAD_INPUT);
    if (dead_) return fail_party("this rank was left out of an earlier round (timed out): open a new net", first_party(rank));
    seq_[sid]++;
    op_[sid] = 0;
    if (transport == ZK_NET_LOCAL) {
      *mask = 1;
      return ZK_OK;
    }
    NetChan& c = ctl_->chan[sid];
    const uint64_t s = seq_[sid];
    c.arrive[rank].store(s, std::memory_order_release);
    relax_reset();
    if (rank == 0) {
      uint64_t dl = net_now_ms() + timeout_ms;
      uint32_t m = 1;
      // (alive_ is shared by the channels, which are driven by different host threads: atomic, and only ever narrowed)
      for (;;) {
        const uint32_t alive = alive_.load(std::memory_order_acquire);
        m = 0;
        for (int r = 0; r < world; r++)
          if ((alive & (1u << r)) && c.arrive[r].load(std::memory_order_acquire) >= s) m |= 1u << r;
        if (m == alive || net_now_ms() > dl) break;
        relax();
      }
      alive_.fetch_and(m, std::memory_order_acq_rel);
      c.verdict_mask.store(m, std::memory_order_relaxed);
      c.verdict_seq.store(s, std::memory_order_release);
      *mask = m;
      return ZK_OK;
    }
    uint64_t dl = net_now_ms() + 2 * timeout_ms + 1000;
    while (c.verdict_seq.load(std::memory_order_acquire) < s) {
      if (net_now_ms() > dl) return fail("the king did not answer (timed out)", ZK_ERR_NOT_CONNECTED);
      relax();
    }
    if (c.verdict_seq.load(std::memory_order_acquire) != s) {
      dead_ = true;
      return fail_party("this rank entered the round after the king's timeout", first_party(rank));
    }
    uint32_t m = c.verdict_mask.load(std::memory_order_relaxed);
    *mask = m;
    if (!(m & (1u << rank))) {
      dead_ = true;
      return fail_party("this rank entered the round after the king's timeout", first_party(rank));
    }
    return ZK_OK;
  }
  uint32_t full_mask() const { return world >= 32 ? 0xffffffffu : ((1u << world) - 1); }
  // pos[p] = index of party p among the parties of the ranks in `mask`, ascending by party id
  void row_positions(uint32_t mask, int* pos) const {
    const int k = parties_per_rank();
    bool here[NET_MAXR * 64] = {};
    for (int r = 0; r < world; r++)
      if (mask & (1u << r))
        for (int i = 0; i < k; i++) here[party(r, i)] = true;
    int c = 0;
    for (int p = 0; p < n; p++) pos[p] = here[p] ? c++ : -1;
  }

  // ---- data plane.  bytes = bytes PER RANK (k parties' rows); the king's `full` holds the present ranks' blocks
  // compacted in rank order (so that a dropout leaves the [np][len] layout the king kernels take).
  // verb counters since creation: gathers, scatters, all-to-alls, payload bytes this rank sent (zk_net_stats)
  std::atomic<uint64_t> stats[4] = {};          // (bumped by the channel threads)

  int gather(int sid, uint32_t mask, const void* local, size_t bytes, void* full) {
    const int op = op_[sid]++;
    stats[0].fetch_add(1, std::memory_order_relaxed);
    if (rank != 0) stats[3].fetch_add(bytes, std::memory_order_relaxed);
    if (transport == ZK_NET_LOCAL) return copy_dd(full, local, bytes, sid);
    const int k = parties_per_rank();
    const size_t rb = bytes / (size_t)k;                 // bytes of one party row (general maps move rows)
    int pos[NET_MAXR * 64];
    if (!contig_) {
      if (bytes % (size_t)k) return fail("with a party_to_rank map the block must be k equal party rows", ZK_ERR_BAD_INPUT);
      row_positions(mask, pos);
    }
    if (transport == ZK_NET_RCCL && !contig_) {
      Rccl& R = Rccl::inst();
      int rc = R.GroupStart();
      for (int r = 0; r < world && !rc; r++) {
        if (!(mask & (1u << r)) || (rank != 0 && r != rank)) continue;
        for (int i = 0; i < k && !rc; i++) {
          const char* src = (const char*)local + (size_t)i * rb;
          if (rank != 0) {
            rc = R.Send(src, rb, 1, 0, comm_[sid], stream_[sid]);
            continue;
          }
          char* dst = (char*)full + (size_t)pos[party(r, i)] * rb;
          if (r == 0) rc = copy_dd(dst, src, rb, sid);
          else rc = R.Recv(dst, rb, 1, r, comm_[sid], stream_[sid]);
        }
      }
      int rc2 = R.GroupEnd();
      if (rc || rc2) return rccl_fail(rc ? rc : rc2, "gather");
      return ZK_OK;
    }
    if (transport == ZK_NET_RCCL) {
      Rccl& R = Rccl::inst();
      int rc = R.GroupStart();
      if (rank == 0) {
        int slot = 0;
        for (int r = 0; r < world && !rc; r++) {
          if (!(mask & (1u << r))) continue;
          char* dst = (char*)full + (size_t)slot * bytes;
          if (r == 0) rc = copy_dd(dst, local, bytes, sid);
          else rc = R.Recv(dst, bytes, /*ncclUint8*/ 1, r, comm_[sid], stream_[sid]);
          slot++;
        }
      } else {
        rc = R.Send(local, bytes, 1, 0, comm_[sid], stream_[sid]);
      }
      int rc2 = R.GroupEnd();
      if (rc || rc2) return rccl_fail(rc ? rc : rc2, "gather");
      return ZK_OK;
    }
    // SHM
    NetChan& c = ctl_->chan[sid];
    for (size_t off = 0, ch = 0; off < bytes || (bytes == 0 && ch == 0); off += cap_, ch++) {
      const size_t len = bytes - off < cap_ ? bytes - off : cap_;
      const uint64_t t = tick(sid, op, ch);
      // my slot is free once the king consumed what I posted last (king_tick is monotonic)
      if (rank != 0 && !wait_ge(c.king_tick, last_g_[sid], "king (gather slot)")) return ZK_ERR_NOT_CONNECTED;
      if (int rc = copy_out(data_[sid] + (size_t)rank * cap_, (const char*)local + off, len, sid)) return rc;
      c.rank_tick[rank].store(t, std::memory_order_release);
      last_g_[sid] = t;
      if (rank == 0) {
        int slot = 0;
        for (int r = 0; r < world; r++) {
          if (!(mask & (1u << r))) continue;
          if (!wait_ge(c.rank_tick[r], t, "peer (gather)")) return ZK_ERR_NOT_CONNECTED;
          if (contig_) {
            if (int rc = copy_in((char*)full + (size_t)slot * bytes + off, data_[sid] + (size_t)r * cap_, len, sid)) return rc;
          } else {
            for (size_t o = off; o < off + len;) {        // the chunk, cut at party-row boundaries
              const size_t row = o / rb, in = o % rb, pl = rb - in < off + len - o ? rb - in : off + len - o;
              if (int rc = copy_in((char*)full + (size_t)pos[party(r, (int)row)] * rb + in, data_[sid] + (size_t)r * cap_ + (o - off), pl, sid))
                return rc;
              o += pl;
            }
          }
          slot++;
        }
        if (int rc = sync(sid)) return rc;
        c.king_tick.store(t, std::memory_order_release);
      }
      if (bytes == 0) break;
    }
    return ZK_OK;
  }

  int scatter(int sid, uint32_t mask, const void* full, size_t bytes, void* local) {
    const int op = op_[sid]++;
    stats[1].fetch_add(1, std::memory_order_relaxed);
    if (rank == 0) stats[3].fetch_add(bytes * (size_t)(__builtin_popcount(mask) - 1), std::memory_order_relaxed);
    if (transport == ZK_NET_LOCAL) return copy_dd(local, full, bytes, sid);
    const int k = parties_per_rank();
    const size_t rb = bytes / (size_t)k;
    if (!contig_ && bytes % (size_t)k) return fail("with a party_to_rank map the block must be k equal party rows", ZK_ERR_BAD_INPUT);
    if (transport == ZK_NET_RCCL && !contig_) {
      Rccl& R = Rccl::inst();
      int rc = R.GroupStart();
      for (int r = 0; r < world && !rc; r++) {
        if (!(mask & (1u << r)) || (rank != 0 && r != rank)) continue;
        for (int i = 0; i < k && !rc; i++) {
          char* dst = (char*)local + (size_t)i * rb;
          if (rank != 0) {
            rc = R.Recv(dst, rb, 1, 0, comm_[sid], stream_[sid]);
            continue;
          }
          const char* src = (const char*)full + (size_t)party(r, i) * rb;     // the king's output is [n][len] by party id
          if (r == 0) rc = copy_dd(dst, src, rb, sid);
          else rc = R.Send(src, rb, 1, r, comm_[sid], stream_[sid]);
        }
      }
      int rc2 = R.GroupEnd();
      if (rc || rc2) return rccl_fail(rc ? rc : rc2, "scatter");
      return ZK_OK;
    }
    if (transport == ZK_NET_RCCL) {
      Rccl& R = Rccl::inst();
      int rc = R.GroupStart();
      if (rank == 0) {
        for (int r = 0; r < world && !rc; r++) {
          if (!(mask & (1u << r))) continue;
          const char* src = (const char*)full + (size_t)r * bytes;        // the king's output is [n][len]: all parties
          if (r == 0) rc = copy_dd(local, src, bytes, sid);
          else rc = R.Send(src, bytes, 1, r, comm_[sid], stream_[sid]);
        }
      } else {
        rc = R.Recv(local, bytes, 1, 0, comm_[sid], stream_[sid]);
      }
      int rc2 = R.GroupEnd();
      if (rc || rc2) return rccl_fail(rc ? rc : rc2, "scatter");
      return ZK_OK;
    }
    NetChan& c = ctl_->chan[sid];
    for (size_t off = 0, ch = 0; off < bytes || (bytes == 0 && ch == 0); off += cap_, ch++) {
      const size_t len = bytes - off < cap_ ? bytes - off : cap_;
      const uint64_t t = tick(sid, op, ch);
      if (rank == 0) {
        for (int r = 1; r < world; r++) {      // every present rank consumed what the king sent it last
          if (!(mask & (1u << r))) continue;
          if (!wait_ge(c.rank_tick[r], last_s_[sid][r], "peer (scatter slot)")) return ZK_ERR_NOT_CONNECTED;
        }
        for (int r = 0; r < world; r++) {
          if (!(mask & (1u << r))) continue;
          if (contig_) {
            if (int rc = copy_out(data_[sid] + (size_t)r * cap_, (const char*)full + (size_t)r * bytes + off, len, sid)) return rc;
          } else {
            for (size_t o = off; o < off + len;) {
              const size_t row = o / rb, in = o % rb, pl = rb - in < off + len - o ? rb - in : off + len - o;
              if (int rc = copy_out_async(data_[sid] + (size_t)r * cap_ + (o - off), (const char*)full + (size_t)party(r, (int)row) * rb + in, pl, sid))
                return rc;
              o += pl;
            }
          }
          last_s_[sid][r] = t;
        }
        if (int rc = sync(sid)) return rc;
        c.king_tick.store(t, std::memory_order_release);
      }
      if (!wait_ge(c.king_tick, t, "king (scatter)")) return ZK_ERR_NOT_CONNECTED;
      if (int rc = copy_in((char*)local + off, data_[sid] + (size_t)rank * cap_, len, sid)) return rc;
      if (int rc = sync(sid)) return rc;
      c.rank_tick[rank].store(t, std::memory_order_release);
      if (bytes == 0) break;
    }
    return ZK_OK;
  }

  // All-to-all among the ranks of `mask` (the second-stage king: every rank is king of a contiguous chunk range).
  //   send : block for rank r at send + r * bytes (indexed by RANK; blocks of absent ranks are ignored)
  //   recv : block from the i-th PRESENT rank at recv + i * bytes (compacted in rank order: the [np][len] layout the king
  //          kernels take after a dropout)
  int alltoall(int sid, uint32_t mask, const void* send, size_t bytes, void* recv) {
    const int op = op_[sid]++;
    stats[2].fetch_add(1, std::memory_order_relaxed);
    stats[3].fetch_add(bytes * (size_t)(__builtin_popcount(mask) - 1), std::memory_order_relaxed);
    if (transport == ZK_NET_LOCAL) return copy_dd(recv, send, bytes, sid);
    int present[NET_MAXR], np_r = 0, me = -1;
    for (int r = 0; r < world; r++)
      if (mask & (1u << r)) {
        if (r == rank) me = np_r;
        present[np_r++] = r;
      }
    if (me < 0) return fail("this rank is not part of the round", ZK_ERR_PROTOCOL);
    if (transport == ZK_NET_RCCL) {
      Rccl& R = Rccl::inst();
      int rc = R.GroupStart();
      for (int i = 0; i < np_r && !rc; i++) {
        const int r = present[i];
        const char* src = (const char*)send + (size_t)r * bytes;
        char* dst = (char*)recv + (size_t)i * bytes;
        if (r == rank) {
          rc = copy_dd(dst, src, bytes, sid);
        } else {
          rc = R.Send(src, bytes, 1, r, comm_[sid], stream_[sid]);
          if (!rc) rc = R.Recv(dst, bytes, 1, r, comm_[sid], stream_[sid]);
        }
      }
      int rc2 = R.GroupEnd();
      if (rc || rc2) return rccl_fail(rc ? rc : rc2, "alltoall");
      return ZK_OK;
    }
    // SHM: my segment slot holds one sub-slot per destination rank
    NetChan& c = ctl_->chan[sid];
    const size_t sub = (cap_ / (size_t)world) & ~(size_t)63;
    if (!sub) return fail("shared segment too small for an all-to-all", ZK_ERR_BAD_INPUT);
    for (size_t off = 0, ch = 0; off < bytes || (bytes == 0 && ch == 0); off += sub, ch++) {
      const size_t len = bytes - off < sub ? bytes - off : sub;
      const uint64_t t = tick(sid, op, ch);
      // my sub-slots are free once every present rank has read what I posted last
      for (int i = 0; i < np_r; i++)
        if (!wait_ge(c.a2a_done[present[i]], last_a_[sid], "peer (all-to-all slot)")) return ZK_ERR_NOT_CONNECTED;
      for (int i = 0; i < np_r; i++) {
        const int r = present[i];
        if (r == rank) continue;
        if (int rc = copy_out_async(adata_[sid] + (size_t)rank * cap_ + (size_t)r * sub, (const char*)send + (size_t)r * bytes + off,
                                    len, sid))
          return rc;
      }
      if (int rc = sync(sid)) return rc;
      c.a2a_post[rank].store(t, std::memory_order_release);
      for (int i = 0; i < np_r; i++) {
        const int r = present[i];
        char* dst = (char*)recv + (size_t)i * bytes + off;
        if (r == rank) {
          if (int rc = copy_dd(dst, (const char*)send + (size_t)r * bytes + off, len, sid)) return rc;
          continue;
        }
        if (!wait_ge(c.a2a_post[r], t, "peer (all-to-all)")) return ZK_ERR_NOT_CONNECTED;
        if (int rc = copy_in(dst, adata_[sid] + (size_t)r * cap_ + (size_t)rank * sub, len, sid)) return rc;
      }
      if (int rc = sync(sid)) return rc;
      c.a2a_done[rank].store(t, std::memory_order_release);
      last_a_[sid] = t;
      if (bytes == 0) break;
    }
    return ZK_OK;
  }

  // small host values: every present rank contributes `bytes` (<= NET_PAYLOAD); the king receives them compacted in
  // rank order in `all` (king only) -- d_msm's "send the masked MSM result to the king" (dmsm/mod.rs:76-84)
  int gather_host(int sid, uint32_t mask, const void* mine, size_t bytes, void* all) {
    const int op = op_[sid]++;
    if (bytes > (size_t)NET_PAYLOAD) return fail("host payload too large", ZK_ERR_BAD_INPUT);
    if (transport == ZK_NET_LOCAL) {
      memcpy(all, mine, bytes);
      return ZK_OK;
    }
    NetChan& c = ctl_->chan[sid];
    const uint64_t t = tick(sid, op, 0);
    memcpy(c.payload[rank], mine, bytes);
    c.pay_tick[rank].store(t, std::memory_order_release);
    if (rank == 0) {
      int slot = 0;
      for (int r = 0; r < world; r++) {
        if (!(mask & (1u << r))) continue;
        if (!wait_ge(c.pay_tick[r], t, "peer (host gather)")) return ZK_ERR_NOT_CONNECTED;
        memcpy((char*)all + (size_t)slot * bytes, c.payload[r], bytes);
        slot++;
      }
    }
    return ZK_OK;
  }
  // the king's answer, the same bytes for everyone (dmsm/mod.rs:87 `king sends the result to all`)
  int bcast_host(int sid, uint32_t mask, void* buf, size_t bytes) {
    const int op = op_[sid]++;
    (void)mask;
    if (bytes > (size_t)NET_PAYLOAD) return fail("host payload too large", ZK_ERR_BAD_INPUT);
    if (transport == ZK_NET_LOCAL) return ZK_OK;
    NetChan& c = ctl_->chan[sid];
    const uint64_t t = tick(sid, op, 0);
    if (rank == 0) {
      memcpy(c.pay_out, buf, bytes);
      c.pay_king.store(t, std::memory_order_release);
    } else {
      if (!wait_ge(c.pay_king, t, "king (host broadcast)")) return ZK_ERR_NOT_CONNECTED;
      memcpy(buf, c.pay_out, bytes);
    }
    return ZK_OK;
  }

  int sync(int sid) {
    if (host_mode || !stream_[sid]) return ZK_OK;
    if (hipStreamSynchronize(stream_[sid]) != hipSuccess) return fail("hipStreamSynchronize");
    return ZK_OK;
  }
  // Watchdog for the RCCL data plane: wait for the channel's stream with a deadline instead of forever; on expiry the
  // communicators are aborted (ncclCommAbort) and the caller gets ZK_ERR_PROTOCOL -- a hung collective must not hang
  // the prover (ser_net.rs:122-125).
  int sync_deadline(int sid) {
    if (host_mode || !stream_[sid]) return ZK_OK;
    uint64_t dl = net_now_ms() + timeout_ms;
    for (;;) {
      hipError_t e = hipStreamQuery(stream_[sid]);
      if (e == hipSuccess) return ZK_OK;
      if (e != hipErrorNotReady) return fail("stream error while waiting for a collective");
      if (net_now_ms() > dl) {
        // abort NOW (ncclCommAbort unblocks the kernels of the hung collective) and refuse every later round
        aborted_ = true;
        dead_ = true;
        if (transport == ZK_NET_RCCL) {
          Rccl& R = Rccl::inst();
          for (int s = 0; s < NET_NSID; s++)
            if (comm_[s] && R.CommAbort) {
              (void)R.CommAbort(comm_[s]);
              comm_[s] = nullptr;
            }
        }
        return fail_party("collective did not complete within the timeout", 0);
      }
      relax();
    }
  }

  // (channels fail independently and from different threads: the message fields are written under a lock)
  int fail(const std::string& m, int code = ZK_ERR_GENERIC) {
    std::lock_guard<std::mutex> lk(err_mu_);
    err = m;
    err_party = -1;
    last_code = code;
    return code;
  }
  int fail_party(const std::string& m, int party) {
    std::lock_guard<std::mutex> lk(err_mu_);
    err = m;
    err_party = party;
    last_code = ZK_ERR_PROTOCOL;
    return ZK_ERR_PROTOCOL;
  }
  std::mutex err_mu_;
  int last_code = ZK_OK;

 private:
  // waits are short when every rank is alive (a few microseconds between neighbours on one node): spin first, then
  // back off to 20 us sleeps so that a long wait (a dead peer, up to the timeout) does not burn a core
  // (the spin counter is per THREAD: the three channels of a rank are driven by three host threads at once,
  // ext_wit.rs:158-170 -- a member counter was a data race between them, found by tests/test_sanitizers.py under TSan)
  static uint32_t& spins() {
    static thread_local uint32_t s = 0;
    return s;
  }
  void relax() {
    if (++spins() < 4000) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      return;
    }
    timespec ts{0, 20000};
    nanosleep(&ts, nullptr);
  }
  void relax_reset() { spins() = 0; }
  uint64_t tick(int sid, int op, size_t chunk) const { return (seq_[sid] << 24) | ((uint64_t)(op & 0xff) << 16) | (chunk & 0xffff); }
  bool wait_ge(std::atomic<uint64_t>& a, uint64_t v, const char* what) {
    // ticks are ordered by (round, operation, chunk)
    uint64_t dl = net_now_ms() + timeout_ms;
    relax_reset();
    while (a.load(std::memory_order_acquire) < v) {
      if (net_now_ms() > dl) {
        fail(std::string("timed out waiting for ") + what, ZK_ERR_NOT_CONNECTED);
        return false;
      }
      relax();
    }
    return true;
  }
  bool map_ctl() {
    int fd = shm_open(name_.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) {
      fail("shm_open failed", ZK_ERR_NOT_CONNECTED);
      return false;
    }
    if (ftruncate(fd, sizeof(NetCtl)) != 0) {
      ::close(fd);
      fail("ftruncate failed", ZK_ERR_NOT_CONNECTED);
      return false;
    }
    void* p = mmap(nullptr, sizeof(NetCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) {
      fail("mmap failed", ZK_ERR_NOT_CONNECTED);
      return false;
    }
    ctl_ = (NetCtl*)p;      // a fresh segment is zero-filled: all sequence words start at 0
    return true;
  }
  bool map_data(int s, char** seg, const char* suffix) {
    std::string nm = name_ + suffix + std::to_string(s);
    int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)(cap_ * world)) != 0) {
      if (fd >= 0) ::close(fd);
      fail("shm data segment", ZK_ERR_NOT_CONNECTED);
      return false;
    }
    void* p = mmap(nullptr, cap_ * world, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) {
      fail("mmap data segment", ZK_ERR_NOT_CONNECTED);
      return false;
    }
    seg[s] = (char*)p;
    return true;
  }
  int copy_dd(void* dst, const void* src, size_t bytes, int sid) {
    if (dst == src || !bytes) return ZK_OK;
    if (host_mode) {
      memmove(dst, src, bytes);
      return ZK_OK;
    }
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream_[sid] ? stream_[sid] : nullptr) != hipSuccess)
      return fail("hipMemcpy d2d");
    return ZK_OK;
  }
  // device (or host) -> shared segment, blocking
  int copy_out(char* shm, const char* src, size_t len, int sid) {
    if (!len) return ZK_OK;
    if (host_mode) {
      memcpy(shm, src, len);
      return ZK_OK;
    }
    if (hipMemcpyAsync(shm, src, len, hipMemcpyDeviceToHost, stream_[sid]) != hipSuccess) return fail("hipMemcpy d2h");
    return sync(sid);
  }
  int copy_out_async(char* shm, const char* src, size_t len, int sid) {     // caller syncs
    if (!len) return ZK_OK;
    if (host_mode) {
      memcpy(shm, src, len);
      return ZK_OK;
    }
    if (hipMemcpyAsync(shm, src, len, hipMemcpyDeviceToHost, stream_[sid]) != hipSuccess) return fail("hipMemcpy d2h");
    return ZK_OK;
  }
  int copy_in(char* dst, const char* shm, size_t len, int sid) {
    if (!len) return ZK_OK;
    if (host_mode) {
      memcpy(dst, shm, len);
      return ZK_OK;
    }
    if (hipMemcpyAsync(dst, shm, len, hipMemcpyHostToDevice, stream_[sid]) != hipSuccess) return fail("hipMemcpy h2d");
    return ZK_OK;
  }
  int rccl_fail(int rc, const char* what) {
    Rccl& R = Rccl::inst();
    return fail(std::string("RCCL ") + what + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "error"));
  }

  std::string name_;
  std::vector<int> slot_party_;           // [world][k] party ids of a non-contiguous map
  bool contig_ = true;
  NetCtl* ctl_ = nullptr;
  char* data_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};
  char* adata_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};     // all-to-all staging (own segment: no slot is shared
                                                                    // with the star verbs)
  uint64_t last_a_[NET_NSID] = {0, 0, 0, 0};
  size_t cap_ = 0;
  void* comm_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_in_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_out_[NET_NSID] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t seq_[NET_NSID] = {0, 0, 0, 0};
  uint64_t last_g_[NET_NSID] = {0, 0, 0, 0};
  uint64_t last_s_[NET_NSID][NET_MAXR] = {};
  int op_[NET_NSID] = {0, 0, 0, 0};
  std::atomic<bool> aborted_{false};
  std::atomic<bool> dead_{false};     // this rank was left out of a round
  std::atomic<uint32_t> alive_{0xffffffffu};   // king: ranks still taking part (narrowed by every round's verdict)
};

}  // namespace zk
