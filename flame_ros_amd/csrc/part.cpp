// flame_ros_amd/csrc/part.cpp -- partition mode in the library: ONE graph cut into
// world x parts_per_rank subdomains, the parts of this rank solved by the ordinary single-GPU handles, their halo
// records exchanged by RCCL -- ncclGroupStart / ncclSend + ncclRecv per neighbouring part / ncclGroupEnd on the solve
// stream -- and the costs reduced by one ncclAllReduce of two doubles.  No Python anywhere: flame::Flame (or any C /
// C++ caller) runs a partitioned solve through include/flame_hip.h's flame_hip_comm_* / flame_hip_part_* functions.
// flame_ros_amd/dist.py (torch.distributed) stays as the test harness of the same scheme; both build the SAME
// subdomains (tests/test_part_host.py compares them array for array), so what the gloo tests prove about the
// scheme holds for this file.
//
// The reference has no counterpart (single process, reference src/flame_offline_tum.cc:403-563); the contract is
// BASELINE.json configs 4 / 5.  Layering: this file uses only the public C ABI of the graph handles -- plus, for the r06 PEER
// transport (two launches per exchange, records written straight into the receivers' inboxes: exchange_peer() below), the two
// transport kernels of kernels.h.
//
// RCCL is loaded with dlopen at the first use (librccl.so.1): the solver library itself keeps no link-time dependency
// on it, and a process that already holds an RCCL (PyTorch) shares that copy.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flame_hip.h"
#include "kernels.h"

using flamehip::HaloPartDev;
using flamehip::HaloSegDev;
using flamehip::HaloXArgs;

namespace {

#define HIPCHK(expr)                                             \
  do {                                                           \
    hipError_t e__ = (expr);                                     \
    if (e__ != hipSuccess) return FLAME_HIP_ERR_HIP - (int)e__;  \
  } while (0)

// ---------------------------------------------------------------- RCCL, resolved at run time
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym))
    SYM(GetUniqueId, ncclGetUniqueId); SYM(CommInitRank, ncclCommInitRank); SYM(CommDestroy, ncclCommDestroy);
    SYM(GroupStart, ncclGroupStart); SYM(GroupEnd, ncclGroupEnd); SYM(Send, ncclSend); SYM(Recv, ncclRecv);
    SYM(AllReduce, ncclAllReduce); SYM(AllGather, ncclAllGather); SYM(CommCount, ncclCommCount);
#undef SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllReduce &&
           r.AllGather && r.CommCount;
  });
  return r;
}

#define NCCLCHK(expr)                                                       \
  do {                                                                      \
    ncclResult_t r__ = (expr);                                              \
    if (r__ != ncclSuccess) return FLAME_HIP_ERR_RCCL - (int)r__;           \
  } while (0)

// ---------------------------------------------------------------- host side: partition, subdomains, messages
// Recursive coordinate bisection into nparts near-equal parts (METIS is not in the image: BASELINE config 4's
// "METIS 2-way cut" is an RCB cut here; planar graph: cut ~ sqrt(V)).  The rule of dist.py rcb_parts: split the
// longer side of the bounding box, order (coordinate, id).
void rcb_parts(const float* pos, int32_t V, int nparts, std::vector<int32_t>* part) {
  part->assign((size_t)V, 0);
  std::vector<int32_t> idx((size_t)V);
  std::iota(idx.begin(), idx.end(), 0);
  struct Job { int32_t lo, hi; int first, n; };
  std::vector<Job> stack{{0, V, 0, nparts}};
  while (!stack.empty()) {
    const Job j = stack.back();
    stack.pop_back();
    const int32_t len = j.hi - j.lo;
    if (j.n == 1 || len <= 1) {
      for (int32_t k = j.lo; k < j.hi; ++k) (*part)[(size_t)idx[(size_t)k]] = j.first;
      continue;
    }
    float mn[2] = {pos[2 * (size_t)idx[(size_t)j.lo]], pos[2 * (size_t)idx[(size_t)j.lo] + 1]}, mx[2] = {mn[0], mn[1]};
    for (int32_t k = j.lo; k < j.hi; ++k)
      for (int a = 0; a < 2; ++a) {
        const float c = pos[2 * (size_t)idx[(size_t)k] + a];
        mn[a] = std::min(mn[a], c); mx[a] = std::max(mx[a], c);
      }
    const int axis = (mx[1] - mn[1]) > (mx[0] - mn[0]) ? 1 : 0;
    const int n1 = j.n / 2;
    const int32_t k = (int32_t)(((int64_t)len * n1) / j.n);
    std::sort(idx.begin() + j.lo, idx.begin() + j.hi, [&](int32_t a, int32_t b) {
      const float ca = pos[2 * (size_t)a + axis], cb = pos[2 * (size_t)b + axis];
      return ca < cb || (ca == cb && a < b);
    });
    // (the children keep the parent's order of ids only through the (coordinate, id) rule of their own sort)
    stack.push_back({j.lo + k, j.hi, j.first + n1, j.n - n1});
    stack.push_back({j.lo, j.lo + k, j.first, n1});
  }
}

struct Csr {
  std::vector<int64_t> ptr;
  std::vector<int32_t> adj;
};

void build_csr(int32_t V, int32_t E, const int32_t* edges, Csr* c) {
  c->ptr.assign((size_t)V + 1, 0);
  for (int32_t e = 0; e < E; ++e) { c->ptr[(size_t)edges[2 * e] + 1]++; c->ptr[(size_t)edges[2 * e + 1] + 1]++; }
  for (int32_t v = 0; v < V; ++v) c->ptr[(size_t)v + 1] += c->ptr[(size_t)v];
  c->adj.resize((size_t)2 * E);
  std::vector<int64_t> cur(c->ptr.begin(), c->ptr.end() - 1);
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = edges[2 * e], j = edges[2 * e + 1];
    c->adj[(size_t)cur[(size_t)i]++] = j;
    c->adj[(size_t)cur[(size_t)j]++] = i;
  }
}

// One subdomain = the own vertices of `part_id` + `depth` halo rings (dist.py build_subdomain, same arrays).
struct Subdomain {
  int part_id = 0;
  int32_t n_own = 0;
  std::vector<int32_t> vid;     // global vertex ids: own (ascending), then halo (ascending)
  std::vector<int32_t> eid;     // global edge ids, ascending (keeps every vertex's sum order)
  std::vector<int32_t> ledges;  // 2 e_loc local vertex ids, orientation preserved
  std::vector<uint8_t> e_owned; // this part owns the edge (= owns its source vertex)
};

void build_subdomain(int32_t V, int32_t E, const int32_t* edges, const std::vector<int32_t>& part, const Csr& csr, int part_id,
                     int depth, std::vector<int32_t>* ring /* scratch, V */, Subdomain* s) {
  s->part_id = part_id;
  const int32_t far = depth + 1;
  ring->assign((size_t)V, far);
  std::vector<int32_t> frontier, next;
  for (int32_t v = 0; v < V; ++v)
    if (part[(size_t)v] == part_id) { (*ring)[(size_t)v] = 0; frontier.push_back(v); }
  s->vid = frontier;
  s->n_own = (int32_t)frontier.size();
  std::vector<int32_t> halo;
  for (int r = 1; r <= depth && !frontier.empty(); ++r) {
    next.clear();
    for (int32_t v : frontier)
      for (int64_t k = csr.ptr[(size_t)v]; k < csr.ptr[(size_t)v + 1]; ++k) {
        const int32_t u = csr.adj[(size_t)k];
        if ((*ring)[(size_t)u] == far) { (*ring)[(size_t)u] = r; next.push_back(u); }
      }
    halo.insert(halo.end(), next.begin(), next.end());
    frontier.swap(next);
  }
  std::sort(halo.begin(), halo.end());
  s->vid.insert(s->vid.end(), halo.begin(), halo.end());
  std::vector<int32_t> lid((size_t)V, -1);
  for (size_t k = 0; k < s->vid.size(); ++k) lid[(size_t)s->vid[k]] = (int32_t)k;
  const int32_t inner = std::max(depth, 1);
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = edges[2 * e], j = edges[2 * e + 1];
    const int32_t ri = (*ring)[(size_t)i], rj = (*ring)[(size_t)j];
    if (ri <= depth && rj <= depth && std::min(ri, rj) < inner) {
      s->eid.push_back(e);
      s->ledges.push_back(lid[(size_t)i]);
      s->ledges.push_back(lid[(size_t)j]);
      s->e_owned.push_back(part[(size_t)i] == part_id ? 1 : 0);
    }
  }
}

// what part `src` sends to part `dst` every exchange: global ids, ascending (both sides derive the same lists)
struct Message {
  int src = 0, dst = 0;
  std::vector<int32_t> v, e;
};

struct LocalPart {
  Subdomain sub;
  std::vector<int> peers;                        // neighbouring parts, ascending
  std::vector<int32_t> send_v, send_e, recv_v, recv_e;  // LOCAL ids, peers in order (the pack / unpack layout)
  std::vector<int32_t> send_cnt, recv_cnt;       // 2 per peer: {vertices, edges}
  flame_hip_graph* g = nullptr;
  float* sbuf = nullptr;                         // device
  float* rbuf = nullptr;
  std::vector<uint8_t> vmask;                    // owned vertices / edges (costs)
};

struct P2P {
  int src, dst, kind;  // kind 0 vertex records, 1 edge records
  bool send;
  float* buf;
  size_t count;        // floats
  int peer_rank;
};

constexpr int kVRec = 6, kERec = 3;  // floats per vertex / edge record (flame_hip_halo_pack)

// what EVERY part (not only this rank's) receives, peers ascending: 2 counts per peer {vertex records, edge records}.  Every
// rank derives it from the whole graph, so a sender knows where its records belong in any receiver's inbox.
struct PartIO {
  std::vector<int> peers;
  std::vector<int32_t> recv_cnt;
};

// ---- peer transport: inboxes.  One block of UNCACHED device memory per rank: [flag words: one per (local part, peer)]
// [record buffer, parity 0: the parts' receive buffers in the pack layout][the same, parity 1]; every piece 256-byte aligned.
constexpr size_t kInboxAlign = 256;
inline size_t align_up(size_t x) { return (x + kInboxAlign - 1) & ~(kInboxAlign - 1); }
struct InboxLayout {
  std::vector<size_t> flag_off;     // per local part: byte offset of its first flag word
  std::vector<size_t> rbuf_off[2];  // per local part, per parity: byte offset of its record buffer
  size_t bytes = 0;
};
// Freed uncached blocks are never handed back to the runtime (flame_hip.cpp UncachedPool: a later ordinary hipMalloc that
// recycles one misbehaves on ROCm 7.2): inboxes wait here for the next partition of the process.
struct InboxPool {
  std::mutex m;
  struct Block { void* p; size_t bytes; int device; };
  std::vector<Block> free_blocks;
};
InboxPool& inbox_pool() { static InboxPool p; return p; }

struct PeerBlob {  // FLAME_HIP_PEER_BLOB_BYTES = 128
  uint64_t magic, pid, ptr, bytes;
  hipIpcMemHandle_t handle;  // 64 bytes
  char pad[128 - 4 * 8 - sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(PeerBlob) == FLAME_HIP_PEER_BLOB_BYTES, "blob layout");
constexpr uint64_t kBlobMagic = 0x464c414d45504552ull;  // "FLAMEPER"

}  // namespace

struct flame_hip_comm {
  int device = -1, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t xstream = nullptr;  // pipelined exchanges (parts_per_rank >= 2): the group of part i travels here while part
                                  // i + 1 iterates on `stream`
  hipEvent_t ev_pack = nullptr, ev_x = nullptr;
  double* red = nullptr;  // device: 2 doubles (cost reduction)
  int32_t* flag = nullptr;  // device: the ranks' "a launch of resident tiles gave up" word (flame_hip_part_sync)
  int rccl_ranks = 0;       // ncclCommCount
  bool local = false;       // flame_hip_comm_create_local: no RCCL (peer transport only)
  bool shared_gpu = false;  // two ranks of this communicator sit on one GPU: their parts solve by launches (resident
                            // tiles assume the whole chip; dist.py / bench.py guard the same case)
  std::vector<flame_hip_part*> parts;  // the parts built on this communicator (destroyed first, or detached)
};

struct flame_hip_part {
  flame_hip_comm* comm = nullptr;  // null: host-only plan (tests)
  int rank = 0, world = 1, k = 1, depth = 0;
  int32_t V = 0, E = 0;
  std::vector<int32_t> part;
  std::vector<LocalPart> parts;
  std::vector<P2P> ops;      // this rank's sends, then its receives, each sorted by (src, dst, kind)
  int rings_left = 0;
  int64_t exchanges = 0;
  int device = -1;
  bool persist = true;       // the parts solve with resident tiles (off when ranks share a GPU)
  // the solves queued since the last synchronising call, and the rings the first of them started on: what
  // flame_hip_part_sync repeats by launches, from the snapshot, when any rank's resident tiles gave up
  struct Call { flame_hip_params p; int32_t n; };
  std::vector<Call> txn;
  int txn_rings = 0;
  int64_t txn_exchanges = 0;  // `exchanges` in front of the queued solves (a roll-back discards the ones since)
  int txn_timed = 0;
  int64_t exchanges_discarded = 0;  // exchanges of solves that were rolled back and repeated (info "exchanges_discarded")
  int64_t recovered = 0;
  // option "time_exchanges": HIP events around every exchange (pack -> group of sends / receives -> unpack) of the solves
  // that follow, up to kMaxTimed of them; info "exchange_ns" = their mean once the stream has been synchronised
  static constexpr int kMaxTimed = 64;
  // option "pipeline" (acts with parts_per_rank >= 2): SURVEY 8e "overlap compute with the exchange" by
  // over-decomposition -- inside a solve call the records of part i leave (their own ncclGroup, on the communicator's second
  // stream) as soon as part i has iterated, while part i + 1 iterates; the unpack waits for the last group.
  // Measured on ONE GPU (every record a send / receive of the rank with itself, profiles/r05_part_pipeline_ab.txt): 2 parts
  // 6.8 -> 9.0 us per iteration (two ncclGroups cost more than the 19 us of a part's iterations hide), 4 parts 22.1 -> 20.6,
  // 8 parts of the 200 k graph 92 -> 67: the default is ON from 4 parts per rank (-1 = that rule; 0 / 1 force it).
  int pipeline = -1;
  int64_t pipelined = 0;  // exchanges that went that way (info "exchanges_pipelined")
  bool time_exchanges = false;
  std::vector<hipEvent_t> tev;  // 2 per timed exchange
  int timed = 0;
  // r06 peer transport (option "transport" 1)
  std::vector<PartIO> io;       // every part of the partition
  int transport = 0;
  struct Peer {
    char* inbox = nullptr;      // this rank's
    bool inbox_uncached = false;  // ... in uncached memory (ranks in other processes / on other GPUs write into it while kernels
                                  // of this rank run); a partition whose ranks all sit in ONE process on one stream (world 1)
                                  // takes ordinary device memory: kernel boundaries order its exchanges, and uncached memory takes
                                  // 16-byte stores at ~57 GB/s (76 us for the 4.3 MB of a 50 k graph cut in two at halo depth 48)
    size_t inbox_bytes = 0;
    std::vector<char*> base;    // every rank's inbox in THIS process' address space (own, same-process pointer, or hipIpc mapping)
    std::vector<void*> opened;  // hipIpcOpenMemHandle mappings to close
    bool connected = false, tables = false;
    HaloSegDev* push_dev = nullptr;
    HaloSegDev* pull_dev = nullptr;
    HaloPartDev* parts_dev = nullptr;
    std::vector<HaloPartDev> parts_host;
    int n_push = 0, n_pull = 0, total_push = 0, total_pull = 0;
    int32_t* words = nullptr;   // device: [0] push counter, [1] pull error
    int32_t epoch = 0;          // exchanges through this transport so far (never rolls back: the flags only grow)
    int64_t timeouts = 0;
  } px;
};

namespace {

int plan_parts(flame_hip_part* P, const float* pos, const int32_t* edges) {
  const int nparts = P->world * P->k;
  rcb_parts(pos, P->V, nparts, &P->part);
  Csr csr;
  build_csr(P->V, P->E, edges, &csr);
  // every part's subdomain is derived here (each rank holds the whole graph): what a remote part wants from a local
  // one follows from ITS halo, so no request lists travel between the ranks.  The parts are independent: up to 8 host
  // threads take them in turn (200 k vertices cut 8-way: 150 ms on one thread).
  std::vector<Subdomain> all((size_t)nparts);
  {
    std::atomic<int> next(0);
    auto work = [&]() {
      std::vector<int32_t> ring;
      for (int p = next.fetch_add(1); p < nparts; p = next.fetch_add(1))
        build_subdomain(P->V, P->E, edges, P->part, csr, p, P->depth, &ring, &all[(size_t)p]);
    };
    std::vector<std::thread> th;
    const int nth = std::min(nparts, 8) - 1;
    for (int t = 0; t < nth; ++t) {
      try { th.emplace_back(work); } catch (...) { break; }  // (no thread to be had: this one does what is left)
    }
    work();
    for (std::thread& t : th) t.join();
  }
  // messages src -> dst: the halo vertices of dst that src owns, the non-owned local edges of dst whose source src owns
  std::vector<std::vector<Message>> to((size_t)nparts);  // [dst] -> messages, src ascending
  for (int d = 0; d < nparts; ++d) {
    const Subdomain& s = all[(size_t)d];
    std::vector<Message> m((size_t)nparts);
    for (size_t k = (size_t)s.n_own; k < s.vid.size(); ++k) m[(size_t)P->part[(size_t)s.vid[k]]].v.push_back(s.vid[k]);
    for (size_t k = 0; k < s.eid.size(); ++k)
      if (!s.e_owned[k]) m[(size_t)P->part[(size_t)edges[2 * (size_t)s.eid[k]]]].e.push_back(s.eid[k]);
    for (int o = 0; o < nparts; ++o)
      if (o != d && (!m[(size_t)o].v.empty() || !m[(size_t)o].e.empty())) {
        m[(size_t)o].src = o; m[(size_t)o].dst = d;
        to[(size_t)d].push_back(std::move(m[(size_t)o]));
      }
  }
  // what every part receives, peers ascending (the peer transport's inbox layout of ANY rank follows from it)
  P->io.assign((size_t)nparts, PartIO());
  {
    std::vector<std::vector<char>> peer((size_t)nparts, std::vector<char>((size_t)nparts, 0));
    for (int d = 0; d < nparts; ++d)
      for (const Message& m : to[(size_t)d]) { peer[(size_t)d][(size_t)m.src] = 1; peer[(size_t)m.src][(size_t)d] = 1; }
    for (int d = 0; d < nparts; ++d)
      for (int p = 0; p < nparts; ++p) {
        if (!peer[(size_t)d][(size_t)p]) continue;
        const Message* in = nullptr;
        for (const Message& m : to[(size_t)d]) if (m.src == p) in = &m;
        P->io[(size_t)d].peers.push_back(p);
        P->io[(size_t)d].recv_cnt.push_back(in ? (int32_t)in->v.size() : 0);
        P->io[(size_t)d].recv_cnt.push_back(in ? (int32_t)in->e.size() : 0);
      }
  }
  P->parts.clear();
  P->parts.resize((size_t)P->k);
  for (int i = 0; i < P->k; ++i) {
    LocalPart& L = P->parts[(size_t)i];
    const int me = P->rank * P->k + i;
    L.sub = std::move(all[(size_t)me]);
    std::vector<int32_t> lid_v((size_t)P->V, -1);
    for (size_t k = 0; k < L.sub.vid.size(); ++k) lid_v[(size_t)L.sub.vid[k]] = (int32_t)k;
    auto lid_e = [&](int32_t ge) { return (int32_t)(std::lower_bound(L.sub.eid.begin(), L.sub.eid.end(), ge) - L.sub.eid.begin()); };
    std::vector<bool> is_peer((size_t)nparts, false);
    for (const Message& m : to[(size_t)me]) is_peer[(size_t)m.src] = true;
    for (int d = 0; d < nparts; ++d)
      for (const Message& m : to[(size_t)d])
        if (m.src == me) is_peer[(size_t)d] = true;
    for (int p = 0; p < nparts; ++p)
      if (is_peer[(size_t)p]) L.peers.push_back(p);
    for (int p : L.peers) {
      const Message* out = nullptr;  // me -> p
      for (const Message& m : to[(size_t)p]) if (m.src == me) out = &m;
      const Message* in = nullptr;   // p -> me
      for (const Message& m : to[(size_t)me]) if (m.src == p) in = &m;
      L.send_cnt.push_back(out ? (int32_t)out->v.size() : 0); L.send_cnt.push_back(out ? (int32_t)out->e.size() : 0);
      L.recv_cnt.push_back(in ? (int32_t)in->v.size() : 0); L.recv_cnt.push_back(in ? (int32_t)in->e.size() : 0);
      if (out) {
        for (int32_t gv : out->v) L.send_v.push_back(lid_v[(size_t)gv]);
        for (int32_t ge : out->e) L.send_e.push_back(lid_e(ge));
      }
      if (in) {
        for (int32_t gv : in->v) L.recv_v.push_back(lid_v[(size_t)gv]);
        for (int32_t ge : in->e) L.recv_e.push_back(lid_e(ge));
      }
    }
    L.vmask.assign(L.sub.vid.size(), 0);
    std::fill(L.vmask.begin(), L.vmask.begin() + L.sub.n_own, 1);
  }
  return 0;
}

// the packed buffer of a part: all vertex records (peers in order), then all edge records (peers in order)
void build_ops(flame_hip_part* P) {
  std::vector<P2P> sends, recvs;
  for (LocalPart& L : P->parts) {
    const int me = L.sub.part_id;
    const size_t nsv = L.send_v.size(), nrv = L.recv_v.size();
    size_t sv = 0, se = kVRec * nsv, rv = 0, re = kVRec * nrv;
    for (size_t i = 0; i < L.peers.size(); ++i) {
      const int p = L.peers[i], pr = p / P->k;
      const size_t a = (size_t)L.send_cnt[2 * i], b = (size_t)L.send_cnt[2 * i + 1];
      const size_t c = (size_t)L.recv_cnt[2 * i], d = (size_t)L.recv_cnt[2 * i + 1];
      if (a) sends.push_back({me, p, 0, true, L.sbuf + sv, kVRec * a, pr});
      if (b) sends.push_back({me, p, 1, true, L.sbuf + se, kERec * b, pr});
      if (c) recvs.push_back({p, me, 0, false, L.rbuf + rv, kVRec * c, pr});
      if (d) recvs.push_back({p, me, 1, false, L.rbuf + re, kERec * d, pr});
      sv += kVRec * a; se += kERec * b; rv += kVRec * c; re += kERec * d;
    }
  }
  // ncclSend / ncclRecv carry no tags: the messages between one pair of ranks match by ORDER, so both sides issue
  // them sorted by (sending part, receiving part, vertex records before edge records)
  auto key = [](const P2P& x, const P2P& y) {
    return x.src != y.src ? x.src < y.src : (x.dst != y.dst ? x.dst < y.dst : x.kind < y.kind);
  };
  std::sort(sends.begin(), sends.end(), key);
  std::sort(recvs.begin(), recvs.end(), key);
  P->ops = sends;
  P->ops.insert(P->ops.end(), recvs.begin(), recvs.end());
}

// ---------------------------------------------------------------- peer transport (r06)
// The inbox of rank r, as every rank computes it (PartIO of its parts).
InboxLayout inbox_layout(const flame_hip_part* P, int r) {
  InboxLayout L;
  size_t off = 0;
  for (int i = 0; i < P->k; ++i) {
    L.flag_off.push_back(off);
    off += sizeof(int32_t) * P->io[(size_t)(r * P->k + i)].peers.size();
  }
  off = align_up(std::max<size_t>(off, 4));
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < P->k; ++i) {
      const PartIO& io = P->io[(size_t)(r * P->k + i)];
      size_t fl = 0;
      for (size_t j = 0; j < io.peers.size(); ++j) fl += (size_t)flamehip::kPeerVRec * (size_t)io.recv_cnt[2 * j] + (size_t)flamehip::kPeerERec * (size_t)io.recv_cnt[2 * j + 1];
      L.rbuf_off[b].push_back(off);
      off += align_up(sizeof(float) * std::max<size_t>(fl, 1));
    }
  L.bytes = off;
  return L;
}

// where the records of message src -> dst (kind 0 vertex / 1 edge) lie inside dst's record buffer, in floats (the pack /
// unpack order: all vertex records, peers in order, then all edge records -- in the peer transport's 32 / 16-byte records),
// and the index of src among dst's peers
bool message_slot(const flame_hip_part* P, int src, int dst, int kind, size_t* off_floats, int* peer_index) {
  const PartIO& io = P->io[(size_t)dst];
  size_t v_before = 0, e_before = 0, v_all = 0;
  int idx = -1;
  for (size_t j = 0; j < io.peers.size(); ++j) {
    if (io.peers[j] == src) idx = (int)j;
    if (idx < 0) { v_before += (size_t)io.recv_cnt[2 * j]; e_before += (size_t)io.recv_cnt[2 * j + 1]; }
    v_all += (size_t)io.recv_cnt[2 * j];
  }
  if (idx < 0) return false;
  *off_floats = kind == 0 ? (size_t)flamehip::kPeerVRec * v_before : (size_t)flamehip::kPeerVRec * v_all + (size_t)flamehip::kPeerERec * e_before;
  *peer_index = idx;
  return true;
}

void peer_release(flame_hip_part* P) {
  flame_hip_part::Peer& X = P->px;
  for (void* p : X.opened) (void)hipIpcCloseMemHandle(p);
  X.opened.clear();
  if (X.push_dev) (void)hipFree(X.push_dev);
  if (X.pull_dev) (void)hipFree(X.pull_dev);
  if (X.parts_dev) (void)hipFree(X.parts_dev);
  if (X.words) (void)hipFree(X.words);
  X.push_dev = X.pull_dev = nullptr; X.parts_dev = nullptr; X.words = nullptr;
  if (X.inbox && X.inbox_uncached) {
    std::lock_guard<std::mutex> lk(inbox_pool().m);
    try { inbox_pool().free_blocks.push_back({X.inbox, X.inbox_bytes, P->device}); } catch (...) {}  // (leaked rather than freed)
  } else if (X.inbox) {
    (void)hipFree(X.inbox);
  }
  X.inbox = nullptr; X.inbox_bytes = 0; X.base.clear(); X.connected = X.tables = false;
}

// this rank's inbox, zeroed (flags 0 = nothing has arrived; epochs start at 1)
int peer_alloc_inbox(flame_hip_part* P) {
  flame_hip_part::Peer& X = P->px;
  if (X.inbox) return 0;
  const size_t want = inbox_layout(P, P->rank).bytes;
  X.inbox_uncached = P->world > 1;
  if (!X.inbox_uncached) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) return FLAME_HIP_ERR_ALLOC;
    HIPCHK(e);
    X.inbox = (char*)p; X.inbox_bytes = want;
  } else {
    std::lock_guard<std::mutex> lk(inbox_pool().m);
    auto& fb = inbox_pool().free_blocks;
    for (size_t i = 0; i < fb.size(); ++i)
      if (fb[i].device == P->device && fb[i].bytes >= want) { X.inbox = (char*)fb[i].p; X.inbox_bytes = fb[i].bytes; fb.erase(fb.begin() + (long)i); break; }
  }
  if (!X.inbox) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, want, hipDeviceMallocUncached);
    if (e == hipErrorOutOfMemory) return FLAME_HIP_ERR_ALLOC;
    HIPCHK(e);
    X.inbox = (char*)p; X.inbox_bytes = want;
  }
  HIPCHK(hipMemsetAsync(X.inbox, 0, X.inbox_bytes, P->comm->stream));
  HIPCHK(hipStreamSynchronize(P->comm->stream));
  try { X.base.assign((size_t)P->world, nullptr); } catch (...) { return FLAME_HIP_ERR_ALLOC; }
  X.base[(size_t)P->rank] = X.inbox;
  if (P->world == 1) X.connected = true;
  return 0;
}

int peer_make_blob(flame_hip_part* P, PeerBlob* b) {
  int rc = peer_alloc_inbox(P);
  if (rc) return rc;
  std::memset(b, 0, sizeof(*b));
  b->magic = kBlobMagic; b->pid = (uint64_t)getpid(); b->ptr = (uint64_t)(uintptr_t)P->px.inbox; b->bytes = P->px.inbox_bytes;
  HIPCHK(hipIpcGetMemHandle(&b->handle, P->px.inbox));
  return 0;
}

int peer_connect(flame_hip_part* P, const PeerBlob* blobs) {
  flame_hip_part::Peer& X = P->px;
  int rc = peer_alloc_inbox(P);
  if (rc) return rc;
  for (int r = 0; r < P->world; ++r) {
    if (r == P->rank) continue;
    const PeerBlob& b = blobs[r];
    if (b.magic != kBlobMagic || b.bytes < inbox_layout(P, r).bytes) return FLAME_HIP_ERR_ARG;
    if (b.pid == (uint64_t)getpid()) {  // a rank of this very process (threads, tests): its pointer is ours
      X.base[(size_t)r] = (char*)(uintptr_t)b.ptr;
    } else {
      void* p = nullptr;
      HIPCHK(hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess));
      try { X.opened.push_back(p); } catch (...) { (void)hipIpcCloseMemHandle(p); return FLAME_HIP_ERR_ALLOC; }
      X.base[(size_t)r] = (char*)p;
    }
  }
  X.connected = true;
  X.tables = false;
  return 0;
}

// the library's own gathering of the blobs, over the communicator's RCCL (collective)
int peer_connect_rccl(flame_hip_part* P) {
  flame_hip_comm* C = P->comm;
  PeerBlob mine;
  int rc = peer_make_blob(P, &mine);
  if (rc) return rc;
  char* dev = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&dev), sizeof(PeerBlob) * (size_t)P->world));
  std::vector<PeerBlob> all((size_t)P->world);
  hipError_t e = hipMemcpyAsync(dev + sizeof(PeerBlob) * (size_t)P->rank, &mine, sizeof(mine), hipMemcpyHostToDevice, C->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = rccl().AllGather(dev + sizeof(PeerBlob) * (size_t)P->rank, dev, sizeof(PeerBlob), ncclChar, C->comm, C->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(all.data(), dev, sizeof(PeerBlob) * (size_t)P->world, hipMemcpyDeviceToHost, C->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
  (void)hipFree(dev);
  if (r != ncclSuccess) return FLAME_HIP_ERR_RCCL - (int)r;
  HIPCHK(e);
  return peer_connect(P, all.data());
}

// the segment tables of the two transport kernels (device), from the parts' registered lists
int peer_build_tables(flame_hip_part* P, const std::vector<flame_hip_halo_view>& views) {
  flame_hip_part::Peer& X = P->px;
  X.tables = false;  // (until everything below has succeeded)
  std::vector<HaloSegDev> push, pull;
  const InboxLayout mine = inbox_layout(P, P->rank);
  std::vector<InboxLayout> lay((size_t)P->world);
  int32_t tp = 0, tl = 0;
  try {
    for (int r = 0; r < P->world; ++r) lay[(size_t)r] = inbox_layout(P, r);
    for (int i = 0; i < P->k; ++i) {
      const LocalPart& L = P->parts[(size_t)i];
      const int me = L.sub.part_id;
      const flame_hip_halo_view& V = views[(size_t)i];
      size_t sv = 0, se = 0, rv = 0, re = 0;  // running offsets into the part's registered lists (peers in order)
      for (size_t j = 0; j < L.peers.size(); ++j) {
        const int p = L.peers[j], pr = p / P->k, pi = p % P->k;
        const int32_t a = L.send_cnt[2 * j], b = L.send_cnt[2 * j + 1], c = L.recv_cnt[2 * j], d = L.recv_cnt[2 * j + 1];
        for (int kind = 0; kind < 2; ++kind) {
          const int32_t n_out = kind == 0 ? a : b, n_in = kind == 0 ? c : d;
          if (n_out > 0) {  // me -> p: into p's inbox on rank pr
            size_t off = 0; int pidx = -1;
            if (!message_slot(P, me, p, kind, &off, &pidx) || !X.base[(size_t)pr]) return FLAME_HIP_ERR_STATE;
            HaloSegDev s;
            s.idx = kind == 0 ? V.send_v + sv : V.send_e + se;
            for (int par = 0; par < 2; ++par) s.buf[par] = reinterpret_cast<float*>(X.base[(size_t)pr] + lay[(size_t)pr].rbuf_off[par][(size_t)pi]) + off;
            s.flag = reinterpret_cast<int32_t*>(X.base[(size_t)pr] + lay[(size_t)pr].flag_off[(size_t)pi]) + pidx;
            s.didx = nullptr; s.dpart = 0;
            if (pr == P->rank) {  // a part of this rank: straight into its state arrays, at ITS registered receive ids
              const LocalPart& D = P->parts[(size_t)pi];
              size_t dv = 0, de = 0;
              for (size_t jj = 0; jj < D.peers.size() && D.peers[jj] != me; ++jj) { dv += (size_t)D.recv_cnt[2 * jj]; de += (size_t)D.recv_cnt[2 * jj + 1]; }
              s.didx = kind == 0 ? views[(size_t)pi].recv_v + dv : views[(size_t)pi].recv_e + de;
              s.dpart = pi;
              s.flag = nullptr;
            }
            s.first = tp; s.count = n_out; s.kind = kind; s.part = i;
            tp += n_out;
            push.push_back(s);
          }
          if (n_in > 0 && pr != P->rank) {  // p -> me from ANOTHER rank: out of this rank's own inbox
            size_t off = 0; int pidx = -1;
            if (!message_slot(P, p, me, kind, &off, &pidx)) return FLAME_HIP_ERR_STATE;
            HaloSegDev s;
            s.idx = kind == 0 ? V.recv_v + rv : V.recv_e + re;
            for (int par = 0; par < 2; ++par) s.buf[par] = reinterpret_cast<float*>(X.inbox + mine.rbuf_off[par][(size_t)i]) + off;
            s.flag = reinterpret_cast<int32_t*>(X.inbox + mine.flag_off[(size_t)i]) + pidx;
            s.didx = nullptr; s.dpart = 0;
            s.first = tl; s.count = n_in; s.kind = kind; s.part = i;
            tl += n_in;
            pull.push_back(s);
          }
        }
        sv += (size_t)a; se += (size_t)b; rv += (size_t)c; re += (size_t)d;
      }
    }
    X.parts_host.assign((size_t)P->k, HaloPartDev());
  } catch (const std::bad_alloc&) { return FLAME_HIP_ERR_ALLOC; }
  for (int i = 0; i < P->k; ++i)
    for (int b = 0; b < 2; ++b) {
      X.parts_host[(size_t)i].A[b] = (float4*)views[(size_t)i].A[b];
      X.parts_host[(size_t)i].B[b] = (float4*)views[(size_t)i].B[b];
      X.parts_host[(size_t)i].q[b] = (float4*)views[(size_t)i].q[b];
    }
  if (X.push_dev) (void)hipFree(X.push_dev);
  if (X.pull_dev) (void)hipFree(X.pull_dev);
  if (X.parts_dev) (void)hipFree(X.parts_dev);
  X.push_dev = X.pull_dev = nullptr; X.parts_dev = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&X.push_dev), sizeof(HaloSegDev) * std::max<size_t>(push.size(), 1)));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&X.pull_dev), sizeof(HaloSegDev) * std::max<size_t>(pull.size(), 1)));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&X.parts_dev), sizeof(HaloPartDev) * (size_t)P->k));
  if (!X.words) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&X.words), 2 * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(X.words, 0, 2 * sizeof(int32_t), P->comm->stream));
  }
  hipStream_t s = P->comm->stream;
  if (!push.empty()) HIPCHK(hipMemcpyAsync(X.push_dev, push.data(), sizeof(HaloSegDev) * push.size(), hipMemcpyHostToDevice, s));
  if (!pull.empty()) HIPCHK(hipMemcpyAsync(X.pull_dev, pull.data(), sizeof(HaloSegDev) * pull.size(), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(X.parts_dev, X.parts_host.data(), sizeof(HaloPartDev) * (size_t)P->k, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // (the host vectors go out of scope)
  X.n_push = (int)push.size(); X.n_pull = (int)pull.size(); X.total_push = tp; X.total_pull = tl;
  X.tables = true;
  return 0;
}

// One exchange through the peer transport: ONE push launch (every record of every local part, straight into the receivers'
// inboxes, then the messages' flags) and ONE pull launch (wait for this rank's incoming flags, unpack) on the solve stream.
int exchange_peer(flame_hip_part* P) {
  flame_hip_comm* C = P->comm;
  flame_hip_part::Peer& X = P->px;
  if (P->k > 32) return FLAME_HIP_ERR_ARG;  // (the kernels take the parts' current buffers as one 32-bit mask)
  if (!X.connected) return FLAME_HIP_ERR_STATE;  // (flame_hip_part_peer_connect has not happened)
  int rc;
  std::vector<flame_hip_halo_view> views;
  try { views.resize((size_t)P->k); } catch (...) { return FLAME_HIP_ERR_ALLOC; }
  uint32_t cur = 0;
  bool moved = !X.tables;
  for (int i = 0; i < P->k; ++i) {
    if ((rc = flame_hip_halo_view_get(P->parts[(size_t)i].g, C->stream, &views[(size_t)i]))) return rc;
    cur |= (uint32_t)(views[(size_t)i].cur & 1) << i;
    if (X.tables && (X.parts_host[(size_t)i].A[0] != views[(size_t)i].A[0] || X.parts_host[(size_t)i].q[1] != views[(size_t)i].q[1])) moved = true;
  }
  if (moved && (rc = peer_build_tables(P, views))) return rc;  // (first exchange, or a part's buffers were re-allocated)
  const bool timed = P->time_exchanges && P->timed < flame_hip_part::kMaxTimed;
  if (timed) {
    while ((int)P->tev.size() < 2 * (P->timed + 1)) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreate(&e));
      try { P->tev.push_back(e); } catch (...) { (void)hipEventDestroy(e); return FLAME_HIP_ERR_ALLOC; }
    }
    HIPCHK(hipEventRecord(P->tev[2 * (size_t)P->timed], C->stream));
  }
  HaloXArgs a;
  a.parts = X.parts_dev; a.cur_mask = cur; a.epoch = ++X.epoch; a.counter = X.words; a.err = X.words + 1;
  a.timeout_ticks = 500000000;  // 5 s: a rank may be a whole exchange ahead of a neighbour that is still planning
  a.segs = X.push_dev; a.nsegs = X.n_push; a.total = X.total_push;
  HIPCHK(flamehip::launch_halo_push(C->stream, a));
  a.segs = X.pull_dev; a.nsegs = X.n_pull; a.total = X.total_pull;
  HIPCHK(flamehip::launch_halo_pull(C->stream, a));
  for (LocalPart& L : P->parts)
    if ((rc = flame_hip_halo_written(L.g))) return rc;
  if (timed) {
    HIPCHK(hipEventRecord(P->tev[2 * (size_t)P->timed + 1], C->stream));
    ++P->timed;
  }
  ++P->exchanges;
  return 0;
}

int exchange(flame_hip_part* P) {
  flame_hip_comm* C = P->comm;
  if (P->world * P->k == 1 || P->ops.empty()) return 0;
  if (P->transport == 1) return exchange_peer(P);
  if (!C->comm) return FLAME_HIP_ERR_NORCCL;  // (a local communicator has the peer transport only)
  int rc;
  const bool timed = P->time_exchanges && P->timed < flame_hip_part::kMaxTimed;
  if (timed) {
    while ((int)P->tev.size() < 2 * (P->timed + 1)) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreate(&e));
      try { P->tev.push_back(e); } catch (...) { (void)hipEventDestroy(e); return FLAME_HIP_ERR_ALLOC; }
    }
    HIPCHK(hipEventRecord(P->tev[2 * (size_t)P->timed], C->stream));
  }
  for (LocalPart& L : P->parts)
    if ((rc = flame_hip_halo_pack(L.g, L.sbuf, C->stream))) return rc;
  Rccl& R = rccl();
  NCCLCHK(R.GroupStart());
  for (const P2P& op : P->ops) {
    const ncclResult_t r = op.send ? R.Send(op.buf, op.count, ncclFloat, op.peer_rank, C->comm, C->stream)
                                   : R.Recv(op.buf, op.count, ncclFloat, op.peer_rank, C->comm, C->stream);
    if (r != ncclSuccess) { (void)R.GroupEnd(); return FLAME_HIP_ERR_RCCL - (int)r; }
  }
  NCCLCHK(R.GroupEnd());
  for (LocalPart& L : P->parts)
    if ((rc = flame_hip_halo_unpack(L.g, L.rbuf, C->stream))) return rc;
  if (timed) {
    HIPCHK(hipEventRecord(P->tev[2 * (size_t)P->timed + 1], C->stream));
    ++P->timed;
  }
  ++P->exchanges;
  return 0;
}

}  // namespace

extern "C" {

int flame_hip_rccl_available(void) { return rccl().ok ? 1 : 0; }

int flame_hip_comm_get_unique_id(char id[FLAME_HIP_COMM_ID_BYTES]) {
  if (!id) return FLAME_HIP_ERR_ARG;
  if (!rccl().ok) return FLAME_HIP_ERR_NORCCL;
  static_assert(sizeof(ncclUniqueId) <= FLAME_HIP_COMM_ID_BYTES, "unique id does not fit");
  ncclUniqueId u;
  NCCLCHK(rccl().GetUniqueId(&u));
  std::memset(id, 0, FLAME_HIP_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int flame_hip_comm_create(flame_hip_comm** out, int device, int rank, int world, const char id[FLAME_HIP_COMM_ID_BYTES]) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return FLAME_HIP_ERR_ARG;
  *out = nullptr;
  if (!rccl().ok) return FLAME_HIP_ERR_NORCCL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return FLAME_HIP_ERR_NODEVICE;
  flame_hip_comm* c = new (std::nothrow) flame_hip_comm();
  if (!c) return FLAME_HIP_ERR_ALLOC;
  c->device = device; c->rank = rank; c->world = world;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->red), 2 * sizeof(double));
  if (e != hipSuccess) { flame_hip_comm_destroy(c); return FLAME_HIP_ERR_HIP - (int)e; }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { c->comm = nullptr; flame_hip_comm_destroy(c); return FLAME_HIP_ERR_RCCL - (int)r; }
  r = rccl().CommCount(c->comm, &c->rccl_ranks);
  if (r != ncclSuccess) { flame_hip_comm_destroy(c); return FLAME_HIP_ERR_RCCL - (int)r; }
  // which GPU every rank sits on: {host name hash, PCI domain:bus:device} gathered over the communicator itself
  {
    long long* ids = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&ids), sizeof(long long) * 2 * (size_t)world);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->flag), sizeof(int32_t));
    std::vector<long long> host(2 * (size_t)world, 0);
    if (e == hipSuccess) {
      char name[256] = {0};
      (void)gethostname(name, sizeof(name) - 1);
      unsigned long long h = 1469598103934665603ull;
      for (const char* q = name; *q; ++q) h = (h ^ (unsigned char)*q) * 1099511628211ull;
      hipDeviceProp_t pr;
      long long pci = device;
      if (hipGetDeviceProperties(&pr, device) == hipSuccess) pci = ((long long)pr.pciDomainID << 32) | ((long long)pr.pciBusID << 8) | pr.pciDeviceID;
      const long long mine[2] = {(long long)h, pci};
      e = hipMemcpyAsync(ids + 2 * rank, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) r = rccl().AllGather(ids + 2 * rank, ids, 2, ncclInt64, c->comm, c->stream);
      if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(host.data(), ids, sizeof(long long) * 2 * (size_t)world, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (ids) (void)hipFree(ids);
    if (r != ncclSuccess) { flame_hip_comm_destroy(c); return FLAME_HIP_ERR_RCCL - (int)r; }
    if (e != hipSuccess) { flame_hip_comm_destroy(c); return FLAME_HIP_ERR_HIP - (int)e; }
    for (int a = 0; a < world && !c->shared_gpu; ++a)
      for (int b = a + 1; b < world; ++b)
        if (host[2 * (size_t)a] == host[2 * (size_t)b] && host[2 * (size_t)a + 1] == host[2 * (size_t)b + 1]) { c->shared_gpu = true; break; }
  }
  *out = c;
  return 0;
}

int flame_hip_comm_info(const flame_hip_comm* c, const char* key, int64_t* value) {
  if (!c || !key || !value) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  if (k == "rank") *value = c->rank;
  else if (k == "world") *value = c->world;
  else if (k == "rccl_ranks") *value = c->rccl_ranks;
  else if (k == "device") *value = c->device;
  else if (k == "shared_gpu") *value = c->shared_gpu ? 1 : 0;
  else return FLAME_HIP_ERR_ARG;
  return 0;
}

// A communicator WITHOUT RCCL: the ranks exchange through the peer transport only (flame_hip_part_peer_blob / _connect).
int flame_hip_comm_create_local(flame_hip_comm** out, int device, int rank, int world) {
  if (!out || world < 1 || rank < 0 || rank >= world) return FLAME_HIP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return FLAME_HIP_ERR_NODEVICE;
  flame_hip_comm* c = new (std::nothrow) flame_hip_comm();
  if (!c) return FLAME_HIP_ERR_ALLOC;
  c->device = device; c->rank = rank; c->world = world; c->local = true;
  c->shared_gpu = world > 1;  // (nothing tells where the other ranks sit, and without an all-reduce the ranks cannot agree on a
                              // give-up of resident tiles: the parts solve by launches)
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->red), 2 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->flag), sizeof(int32_t));
  if (e != hipSuccess) { flame_hip_comm_destroy(c); return FLAME_HIP_ERR_HIP - (int)e; }
  c->rccl_ranks = 0;
  *out = c;
  return 0;
}

void flame_hip_comm_destroy(flame_hip_comm* c) {
  if (!c) return;
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  // (a part keeps a pointer to its communicator.  Parts that outlive it are detached here -- they can still be
  // destroyed, nothing else: every other entry point refuses a part without a communicator)
  for (flame_hip_part* P : c->parts) P->comm = nullptr;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->red) (void)hipFree(c->red);
  if (c->flag) (void)hipFree(c->flag);
  if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
  if (c->ev_x) (void)hipEventDestroy(c->ev_x);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void* flame_hip_comm_stream(flame_hip_comm* c) { return c ? (void*)c->stream : nullptr; }

void flame_hip_part_destroy(flame_hip_part* P) {
  if (!P) return;
  if (P->device >= 0) (void)hipSetDevice(P->device);
  if (P->comm) {
    (void)hipStreamSynchronize(P->comm->stream);
    auto& v = P->comm->parts;
    v.erase(std::remove(v.begin(), v.end(), P), v.end());
  }
  for (LocalPart& L : P->parts) {
    if (L.g) flame_hip_graph_destroy(L.g);
    if (L.sbuf) (void)hipFree(L.sbuf);
    if (L.rbuf) (void)hipFree(L.rbuf);
  }
  for (hipEvent_t e : P->tev) (void)hipEventDestroy(e);
  peer_release(P);
  delete P;
}

int flame_hip_part_create(flame_hip_part** out, flame_hip_comm* comm, int32_t plan_rank, int32_t plan_world,
                          int32_t parts_per_rank, int32_t halo_depth, int32_t V, int32_t E, const float* pos,
                          const int32_t* edges, const float* alpha, const float* beta, const float* z, const float* wgt,
                          const float* x0) {
  if (!out) return FLAME_HIP_ERR_ARG;
  *out = nullptr;
  if (parts_per_rank < 1 || halo_depth < 1 || halo_depth > 64 || V < 1 || E < 0 || !pos || (E > 0 && !edges)) return FLAME_HIP_ERR_ARG;
  if (comm && (!z || !wgt || (E > 0 && (!alpha || !beta)))) return FLAME_HIP_ERR_ARG;  // (z / wgt are read whatever E is)
  for (int32_t e = 0; e < E; ++e)
    if (edges[2 * e] < 0 || edges[2 * e] >= V || edges[2 * e + 1] < 0 || edges[2 * e + 1] >= V) return FLAME_HIP_ERR_ARG;
  flame_hip_part* P = new (std::nothrow) flame_hip_part();
  if (!P) return FLAME_HIP_ERR_ALLOC;
  P->comm = comm;
  P->rank = comm ? comm->rank : plan_rank;
  P->world = comm ? comm->world : plan_world;
  P->k = parts_per_rank; P->depth = halo_depth; P->V = V; P->E = E;
  if (P->world < 1 || P->rank < 0 || P->rank >= P->world || (int64_t)P->world * P->k > V) { delete P; return FLAME_HIP_ERR_ARG; }
  int rc = plan_parts(P, pos, edges);
  if (rc) { flame_hip_part_destroy(P); return rc; }
  if (comm) {
    P->device = comm->device;
    P->persist = !comm->shared_gpu;
    try { comm->parts.push_back(P); } catch (...) { P->comm = nullptr; flame_hip_part_destroy(P); return FLAME_HIP_ERR_ALLOC; }
    if (hipSetDevice(comm->device) != hipSuccess) { flame_hip_part_destroy(P); return FLAME_HIP_ERR_NODEVICE; }
    for (LocalPart& L : P->parts) {
      const Subdomain& s = L.sub;
      const int32_t nv = (int32_t)s.vid.size(), ne = (int32_t)s.eid.size();
      std::vector<float> lp(2 * (size_t)nv), lz((size_t)nv), lw((size_t)nv), lx, la((size_t)ne), lb((size_t)ne);
      if (x0) lx.resize((size_t)nv);
      for (int32_t k = 0; k < nv; ++k) {
        const size_t gv = (size_t)s.vid[(size_t)k];
        lp[2 * (size_t)k] = pos[2 * gv]; lp[2 * (size_t)k + 1] = pos[2 * gv + 1];
        lz[(size_t)k] = z[gv]; lw[(size_t)k] = wgt[gv];
        if (x0) lx[(size_t)k] = x0[gv];
      }
      for (int32_t k = 0; k < ne; ++k) { la[(size_t)k] = alpha[(size_t)s.eid[(size_t)k]]; lb[(size_t)k] = beta[(size_t)s.eid[(size_t)k]]; }
      if ((rc = flame_hip_graph_create(&L.g, comm->device, nv, ne, 0)) ||
          (rc = flame_hip_set_option(L.g, "persist", P->persist ? 1 : 0)) ||
          (rc = flame_hip_graph_upload(L.g, lp.data(), s.ledges.data(), la.data(), lb.data(), lz.data(), lw.data(),
                                       x0 ? lx.data() : nullptr, nullptr)) ||
          (rc = flame_hip_halo_register(L.g, (int32_t)L.send_v.size(), L.send_v.data(), (int32_t)L.send_e.size(), L.send_e.data(),
                                        (int32_t)L.recv_v.size(), L.recv_v.data(), (int32_t)L.recv_e.size(), L.recv_e.data()))) {
        flame_hip_part_destroy(P);
        return rc;
      }
      const size_t sb = kVRec * L.send_v.size() + kERec * L.send_e.size(), rb = kVRec * L.recv_v.size() + kERec * L.recv_e.size();
      if (hipMalloc(reinterpret_cast<void**>(&L.sbuf), sizeof(float) * std::max<size_t>(sb, 1)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&L.rbuf), sizeof(float) * std::max<size_t>(rb, 1)) != hipSuccess) {
        flame_hip_part_destroy(P);
        return FLAME_HIP_ERR_ALLOC;
      }
    }
    build_ops(P);
  }
  P->rings_left = P->depth;  // a fresh upload holds exact state on every ring
  *out = P;
  return 0;
}

// New frame on the UNCHANGED topology (flame_hip_graph_update_data of every part): data terms, data weights and the
// initial x of the whole graph (caller's order; x0 may be NULL = z).  The state restarts exact on every ring.
int flame_hip_part_update_data(flame_hip_part* P, const float* z, const float* wgt, const float* x0) {
  if (!P || !P->comm || !z || !wgt) return FLAME_HIP_ERR_ARG;
  int rc0;
  if ((rc0 = flame_hip_part_sync(P))) return rc0;
  for (LocalPart& L : P->parts) {
    const size_t nv = L.sub.vid.size();
    std::vector<float> lz(nv), lw(nv), lx;
    if (x0) lx.resize(nv);
    for (size_t k = 0; k < nv; ++k) {
      const size_t gv = (size_t)L.sub.vid[k];
      lz[k] = z[gv]; lw[k] = wgt[gv];
      if (x0) lx[k] = x0[gv];
    }
    const int rc = flame_hip_graph_update_data(L.g, lz.data(), lw.data(), x0 ? lx.data() : nullptr);
    if (rc) return rc;
  }
  P->rings_left = P->depth;
  return 0;
}

// Every local iteration invalidates one halo ring; an exchange (the owners' exact state) makes all `depth` rings
// valid again.  Successive calls continue on whatever rings the previous one left.  Everything is enqueued on the
// communicator's stream: no host synchronisation inside.
// the ncclGroup of ONE sending phase: every message whose SOURCE part has local index `phase` on its rank (this rank's sends
// of that part; its receives from parts of that index anywhere).  Both ends of a pair of ranks derive the same sets in the
// same (src, dst, kind) order, and the phases follow each other in the same order everywhere: messages match by order.
static int exchange_phase(flame_hip_part* P, int phase, hipStream_t s) {
  flame_hip_comm* C = P->comm;
  Rccl& R = rccl();
  bool any = false;
  for (const P2P& op : P->ops) any = any || (op.src % P->k) == phase;
  if (!any) return 0;
  NCCLCHK(R.GroupStart());
  for (const P2P& op : P->ops) {
    if ((op.src % P->k) != phase) continue;
    const ncclResult_t r = op.send ? R.Send(op.buf, op.count, ncclFloat, op.peer_rank, C->comm, s)
                                   : R.Recv(op.buf, op.count, ncclFloat, op.peer_rank, C->comm, s);
    if (r != ncclSuccess) { (void)R.GroupEnd(); return FLAME_HIP_ERR_RCCL - (int)r; }
  }
  NCCLCHK(R.GroupEnd());
  return 0;
}

// (the iterations themselves: shared by flame_hip_part_solve and the repeat of a give-up)
static int run_iterations(flame_hip_part* P, const flame_hip_params* p, int32_t num_iters) {
  int rc;
  flame_hip_comm* C = P->comm;
  for (int32_t done = 0; done < num_iters;) {
    if (P->rings_left == 0) {
      if ((rc = exchange(P))) return rc;
      P->rings_left = P->depth;
    }
    const int32_t n = std::min<int32_t>(P->rings_left, num_iters - done);
    // this chunk uses the rings up and the call goes on: the exchange behind it is certain -- pipeline it
    const bool pipe = P->transport == 0 && (P->pipeline < 0 ? P->k >= 4 : P->pipeline != 0) && P->k >= 2 && !P->ops.empty() && P->rings_left == n && done + n < num_iters && !P->time_exchanges;
    for (size_t i = 0; i < P->parts.size(); ++i) {
      LocalPart& L = P->parts[i];
      if ((rc = flame_hip_solve(L.g, p, n, C->stream))) return rc;
      if (pipe) {
        if ((rc = flame_hip_halo_pack(L.g, L.sbuf, C->stream))) return rc;
        HIPCHK(hipEventRecord(C->ev_pack, C->stream));
        HIPCHK(hipStreamWaitEvent(C->xstream, C->ev_pack, 0));
        if ((rc = exchange_phase(P, (int)i, C->xstream))) return rc;
      }
    }
    P->rings_left -= n;
    done += n;
    if (pipe) {
      HIPCHK(hipEventRecord(C->ev_x, C->xstream));
      HIPCHK(hipStreamWaitEvent(C->stream, C->ev_x, 0));
      for (LocalPart& L : P->parts)
        if ((rc = flame_hip_halo_unpack(L.g, L.rbuf, C->stream))) return rc;
      ++P->exchanges;
      ++P->pipelined;
      P->rings_left = P->depth;
    }
  }
  return 0;
}

int flame_hip_part_solve(flame_hip_part* P, const flame_hip_params* p, int32_t num_iters) {
  if (!P || !P->comm || !p || num_iters < 0) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(P->comm->device));
  int rc;
  if (P->txn.empty() && P->persist) {  // the first solve since a synchronising call: what a give-up rolls back to
    for (LocalPart& L : P->parts)
      if ((rc = flame_hip_state_snapshot(L.g, P->comm->stream))) return rc;
    P->txn_rings = P->rings_left;
    P->txn_exchanges = P->exchanges;
    P->txn_timed = P->timed;
  }
  if (P->persist) {
    try { P->txn.push_back({*p, num_iters}); } catch (...) { return FLAME_HIP_ERR_ALLOC; }
  }
  return run_iterations(P, p, num_iters);
}

// Resident tiles that gave up (a wait of theirs timed out: a foreign kernel held CUs) leave an unfinished solve behind,
// and by now its records have travelled to the peers.  The ranks agree on it (one 4-byte all-reduce), every rank rolls
// its parts back to the snapshot and repeats the queued solves by ordinary launches.
int flame_hip_part_sync(flame_hip_part* P) {
  if (!P || !P->comm) return FLAME_HIP_ERR_ARG;
  flame_hip_comm* C = P->comm;
  HIPCHK(hipSetDevice(C->device));
  HIPCHK(hipStreamSynchronize(C->stream));
  int rc;
  int32_t bad = 0;
  for (LocalPart& L : P->parts) {
    int32_t one = 0;
    if ((rc = flame_hip_persist_take_error(L.g, &one))) return rc;
    bad |= one;
  }
  // (nothing queued since the last call: no rank has anything to agree on.  flame_hip_part_solve / _sync are COLLECTIVE: every
  // rank of the communicator makes the same calls in the same order -- the exchanges inside a solve already require it -- so
  // the queue is empty on all ranks or on none; a rank whose solve returned an error must not go on using the partition)
  if (P->px.words && P->px.epoch > 0) {  // peer transport: did a pull give up waiting for a neighbour's records?
    int32_t w[2] = {0, 0};
    HIPCHK(hipMemcpy(w, P->px.words, sizeof(w), hipMemcpyDeviceToHost));
    if (w[1]) {
      ++P->px.timeouts;
      HIPCHK(hipMemset(P->px.words + 1, 0, sizeof(int32_t)));
      return FLAME_HIP_ERR_STATE;  // (a neighbour rank is gone or minutes behind: the state holds records that never arrived)
    }
  }
  if (P->txn.empty()) return 0;
  if (C->world > 1 && C->comm) {
    HIPCHK(hipMemcpyAsync(C->flag, &bad, sizeof(bad), hipMemcpyHostToDevice, C->stream));
    NCCLCHK(rccl().AllReduce(C->flag, C->flag, 1, ncclInt32, ncclMax, C->comm, C->stream));
    HIPCHK(hipMemcpyAsync(&bad, C->flag, sizeof(bad), hipMemcpyDeviceToHost, C->stream));
    HIPCHK(hipStreamSynchronize(C->stream));
  }
  std::vector<flame_hip_part::Call> calls;
  calls.swap(P->txn);
  if (!bad) return 0;
  std::vector<int64_t> was(P->parts.size(), 1);  // (the parts' "persist" option as the caller left it: restored below)
  for (size_t i = 0; i < P->parts.size(); ++i) {
    LocalPart& L = P->parts[i];
    if ((rc = flame_hip_get_info(L.g, "persist", &was[i])) || (rc = flame_hip_state_rollback(L.g, C->stream)) ||
        (rc = flame_hip_set_option(L.g, "persist", 0)))
      return rc;
  }
  P->rings_left = P->txn_rings;
  P->exchanges_discarded += P->exchanges - P->txn_exchanges;  // (the discarded exchanges are not counted twice)
  P->exchanges = P->txn_exchanges;
  P->timed = P->txn_timed;
  for (const flame_hip_part::Call& c : calls)
    if ((rc = run_iterations(P, &c.p, c.n))) return rc;
  HIPCHK(hipStreamSynchronize(C->stream));
  for (size_t i = 0; i < P->parts.size(); ++i) {
    int32_t one = 0;
    if ((rc = flame_hip_persist_take_error(P->parts[i].g, &one)) || (rc = flame_hip_set_option(P->parts[i].g, "persist", (int32_t)was[i]))) return rc;
  }
  P->recovered += (int64_t)calls.size();
  return 0;
}

// nltgv2_total_smoothness_cost / nltgv2_total_data_cost of the WHOLE graph (reference src/utils.cc:131-136): every part
// sums what it owns, one ncclAllReduce of 2 doubles (SURVEY.md 8e "final cost reduction")
int flame_hip_part_costs(flame_hip_part* P, const flame_hip_params* p, double* smooth, double* data) {
  if (!P || !P->comm || !p) return FLAME_HIP_ERR_ARG;
  flame_hip_comm* C = P->comm;
  HIPCHK(hipSetDevice(C->device));
  int rc;
  if ((rc = flame_hip_part_sync(P))) return rc;  // (first: a give-up is repeated from the snapshot, which an exchange queued
                                                  // in front of it would not survive)
  if (P->rings_left == 0 && P->depth > 0 && P->world * P->k > 1) {  // an owned edge reads its target in ring 1
    if ((rc = exchange(P))) return rc;
    P->rings_left = P->depth;
    HIPCHK(hipStreamSynchronize(C->stream));
  }
  double acc[2] = {0.0, 0.0};
  for (LocalPart& L : P->parts) {
    double s = 0.0, d = 0.0;
    if ((rc = flame_hip_costs_masked(L.g, p, L.vmask.data(), L.sub.e_owned.data(), &s, &d))) return rc;
    acc[0] += s; acc[1] += d;
  }
  if (C->comm) {  // (a local communicator: this rank's owned sums, the application adds the ranks')
    HIPCHK(hipMemcpyAsync(C->red, acc, sizeof(acc), hipMemcpyHostToDevice, C->stream));
    NCCLCHK(rccl().AllReduce(C->red, C->red, 2, ncclDouble, ncclSum, C->comm, C->stream));
    HIPCHK(hipMemcpyAsync(acc, C->red, sizeof(acc), hipMemcpyDeviceToHost, C->stream));
    HIPCHK(hipStreamSynchronize(C->stream));
  }
  if (smooth) *smooth = acc[0];
  if (data) *data = acc[1];
  return 0;
}

// The solution of the WHOLE graph on every rank (x, w1, w2: V floats; q: 3E floats, interleaved; any may be NULL): every
// rank fills in what its parts own, the rest zero, and ONE ncclAllReduce (integer sum of the bit patterns: exact, no
// -0.0 + 0.0) puts the pieces together.  A verification / results call, not part of the iteration.
int flame_hip_part_gather(flame_hip_part* P, float* x, float* w1, float* w2, float* q) {
  if (!P || !P->comm) return FLAME_HIP_ERR_ARG;
  flame_hip_comm* C = P->comm;
  int rc;
  if ((rc = flame_hip_part_sync(P))) return rc;
  const size_t V = (size_t)P->V, E = (size_t)P->E, total = 3 * V + 3 * E;
  std::vector<float> all(total, 0.0f);
  for (LocalPart& L : P->parts) {
    const Subdomain& s = L.sub;
    const size_t nv = s.vid.size(), ne = s.eid.size();
    std::vector<float> lx(nv), l1(nv), l2(nv), lq(3 * ne);
    if ((rc = flame_hip_download(L.g, lx.data(), l1.data(), l2.data(), lq.data()))) return rc;
    for (size_t k = 0; k < (size_t)s.n_own; ++k) {
      const size_t gv = (size_t)s.vid[k];
      all[gv] = lx[k]; all[V + gv] = l1[k]; all[2 * V + gv] = l2[k];
    }
    for (size_t k = 0; k < ne; ++k)
      if (s.e_owned[k]) std::memcpy(&all[3 * V + 3 * (size_t)s.eid[k]], &lq[3 * k], 3 * sizeof(float));
  }
  if (C->world > 1 && C->comm) {  // (a local communicator: what this rank owns, zero elsewhere)
    int32_t* dev = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dev), total * sizeof(int32_t)));
    hipError_t e = hipMemcpyAsync(dev, all.data(), total * sizeof(float), hipMemcpyHostToDevice, C->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = rccl().AllReduce(dev, dev, total, ncclInt32, ncclSum, C->comm, C->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(all.data(), dev, total * sizeof(float), hipMemcpyDeviceToHost, C->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(C->stream);
    (void)hipFree(dev);
    if (r != ncclSuccess) return FLAME_HIP_ERR_RCCL - (int)r;
    HIPCHK(e);
  }
  if (x) std::memcpy(x, &all[0], V * sizeof(float));
  if (w1) std::memcpy(w1, &all[V], V * sizeof(float));
  if (w2) std::memcpy(w2, &all[2 * V], V * sizeof(float));
  if (q && E) std::memcpy(q, &all[3 * V], 3 * E * sizeof(float));
  return 0;
}

// "time_exchanges" (0 / 1): see flame_hip_part_info "exchange_ns"; setting it (either way) forgets the exchanges timed so far
int flame_hip_part_set_option(flame_hip_part* P, const char* key, int32_t value) {
  if (!P || !key) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  if (k == "time_exchanges") { P->time_exchanges = value != 0; P->timed = 0; return 0; }
  if (k == "pipeline") { P->pipeline = value < 0 ? -1 : (value != 0 ? 1 : 0); return 0; }
  if (k == "transport") {  // 0 RCCL (the contract's path), 1 peer (include/flame_hip.h); COLLECTIVE when the ranks are connected over RCCL
    if (!P->comm || (value != 0 && value != 1)) return FLAME_HIP_ERR_ARG;
    if (value == 0 && !P->comm->comm && P->world * P->k > 1) return FLAME_HIP_ERR_NORCCL;  // (a local communicator has no RCCL to go back to)
    if (value == 1 && !P->px.connected) {
      HIPCHK(hipSetDevice(P->comm->device));
      int rc = P->world == 1 ? peer_alloc_inbox(P) : (P->comm->comm ? peer_connect_rccl(P) : peer_alloc_inbox(P));
      if (rc) return rc;  // (local communicator, world > 1: connected by flame_hip_part_peer_connect)
    }
    P->transport = value;
    return 0;
  }
  return FLAME_HIP_ERR_ARG;
}

int flame_hip_part_peer_blob(flame_hip_part* P, char blob[FLAME_HIP_PEER_BLOB_BYTES]) {
  if (!P || !P->comm || !blob) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(P->comm->device));
  PeerBlob b;
  const int rc = peer_make_blob(P, &b);
  if (rc) return rc;
  std::memcpy(blob, &b, sizeof(b));
  return 0;
}

int flame_hip_part_peer_connect(flame_hip_part* P, const char* blobs) {
  if (!P || !P->comm || !blobs) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(P->comm->device));
  std::vector<PeerBlob> all;
  try { all.resize((size_t)P->world); } catch (...) { return FLAME_HIP_ERR_ALLOC; }
  std::memcpy(all.data(), blobs, sizeof(PeerBlob) * (size_t)P->world);
  return peer_connect(P, all.data());
}

// Introspection (tests, bench): scalars and arrays of the plan.  local_part in [0, parts_per_rank).
int flame_hip_part_info(const flame_hip_part* P, const char* key, int32_t local_part, int64_t* value) {
  if (!P || !key || !value) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  if (k == "num_parts") { *value = (int64_t)P->world * P->k; return 0; }
  if (k == "transport") { *value = P->transport; return 0; }
  if (k == "peer_connected") { *value = P->px.connected ? 1 : 0; return 0; }
  if (k == "peer_epoch") { *value = P->px.epoch; return 0; }
  if (k == "peer_timeouts") { *value = P->px.timeouts; return 0; }
  if (k == "inbox_bytes") { *value = (int64_t)P->px.inbox_bytes; return 0; }
  if (k == "parts_per_rank") { *value = P->k; return 0; }
  if (k == "exchanges") { *value = P->exchanges; return 0; }
  if (k == "exchanges_discarded") { *value = P->exchanges_discarded; return 0; }
  if (k == "p2p_ops") { *value = (int64_t)P->ops.size(); return 0; }
  if (k == "rings_left") { *value = P->rings_left; return 0; }
  if (k == "recovered") { *value = P->recovered; return 0; }
  if (k == "exchanges_timed") { *value = P->timed; return 0; }
  if (k == "exchanges_pipelined") { *value = P->pipelined; return 0; }
  if (k == "exchange_ns") {  // mean device time of the timed exchanges (the caller has synchronised: flame_hip_part_sync)
    double sum = 0.0;
    for (int i = 0; i < P->timed; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, P->tev[2 * (size_t)i], P->tev[2 * (size_t)i + 1]) != hipSuccess) return FLAME_HIP_ERR_STATE;
      sum += ms;
    }
    *value = P->timed > 0 ? (int64_t)(sum / P->timed * 1e6) : 0;
    return 0;
  }
  if (k == "persist") { *value = P->persist ? 1 : 0; return 0; }
  if (local_part < 0 || local_part >= P->k) return FLAME_HIP_ERR_ARG;
  const LocalPart& L = P->parts[(size_t)local_part];
  if (k == "part_id") *value = L.sub.part_id;
  else if (k == "n_own") *value = L.sub.n_own;
  else if (k == "n_ext") *value = (int64_t)L.sub.vid.size();
  else if (k == "e_loc") *value = (int64_t)L.sub.eid.size();
  else if (k == "num_peers") *value = (int64_t)L.peers.size();
  else if (k == "send_bytes") *value = 4 * (int64_t)(kVRec * L.send_v.size() + kERec * L.send_e.size());
  else if (k == "recv_bytes") *value = 4 * (int64_t)(kVRec * L.recv_v.size() + kERec * L.recv_e.size());
  else return L.g ? flame_hip_get_info(L.g, key, value) : FLAME_HIP_ERR_ARG;  // ("persist_used", "persist_launches", "num_tiles", "tile_depth", ...: the part's own handle)
  return 0;
}

// int32 arrays: "part" (V), "vid", "eid", "edges" (2 e_loc), "e_owned" (as int32), "peers", "send_v", "send_e", "recv_v",
// "recv_e", "send_cnt", "recv_cnt" (2 per peer).  Returns the element count (copies min(count, cap)), or an error.
int64_t flame_hip_part_array(const flame_hip_part* P, const char* key, int32_t local_part, int32_t* out, int64_t cap) {
  if (!P || !key) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  std::vector<int32_t> tmp;
  const std::vector<int32_t>* src = nullptr;
  if (k == "part") src = &P->part;
  else {
    if (local_part < 0 || local_part >= P->k) return FLAME_HIP_ERR_ARG;
    const LocalPart& L = P->parts[(size_t)local_part];
    if (k == "vid") src = &L.sub.vid;
    else if (k == "eid") src = &L.sub.eid;
    else if (k == "edges") src = &L.sub.ledges;
    else if (k == "send_v") src = &L.send_v;
    else if (k == "send_e") src = &L.send_e;
    else if (k == "recv_v") src = &L.recv_v;
    else if (k == "recv_e") src = &L.recv_e;
    else if (k == "send_cnt") src = &L.send_cnt;
    else if (k == "recv_cnt") src = &L.recv_cnt;
    else if (k == "peers") { tmp.assign(L.peers.begin(), L.peers.end()); src = &tmp; }
    else if (k == "e_owned") { tmp.assign(L.sub.e_owned.begin(), L.sub.e_owned.end()); src = &tmp; }
    else return FLAME_HIP_ERR_ARG;
  }
  const int64_t n = (int64_t)src->size();
  if (out && cap > 0) std::memcpy(out, src->data(), sizeof(int32_t) * (size_t)std::min(n, cap));
  return n;
}

}  // extern "C"
