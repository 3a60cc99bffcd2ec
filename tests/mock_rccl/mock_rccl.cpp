// Test double of the nine librccl entry points the slab step binds (niwqg_amd/csrc/nq_lib.hip: RcclApi), for ranks that are
// THREADS of one process on one GPU: a one-GPU box cannot host a real RCCL communicator of more than one rank, but the
// library's RCCL link -- grouped ncclSend/ncclRecv per row chunk on the exchange stream, ncclAllReduce of the budget sums,
// the event choreography around them -- can still be driven by P concurrent host threads exactly as P processes would.
// Semantics kept from NCCL: calls are matched per (sender, receiver) pair in posting order; a send completes on the
// sender's stream only after the receiver's copy; counts of a matched pair must agree (else abort with a message);
// ncclCommInitRank and ncclAllReduce are collective over the ranks of the communicator.  Loaded through
// NIWQG_AMD_RCCL_LIB; test infrastructure only.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Msg {
  const void* src;
  size_t bytes;
  hipEvent_t ready, done;
  bool consumed = false;
};
struct World {
  int nranks = 0, joined = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg*>> box;      // (from, to) -> posted sends
  // all-reduce rendezvous
  int arrived = 0, generation = 0;
  std::vector<std::vector<double>> contrib;
  std::vector<double> sum;
  long long n_send = 0, n_recv = 0, n_allreduce = 0, bytes = 0;
};
struct Comm {
  World* w;
  int rank;
};
struct Op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
  Msg* msg;
};
std::mutex g_mu;
std::map<unsigned long long, World*> g_worlds;
unsigned long long g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

void die(const char* what) {
  fprintf(stderr, "mock_rccl: %s\n", what);
  abort();
}
#define HCK(x)                                    \
  do {                                            \
    if ((x) != hipSuccess) die(#x " failed");     \
  } while (0)

int flush() {
  // 1. post every send of the group
  for (Op& o : t_ops) {
    if (!o.send) continue;
    Msg* m = new Msg();
    m->src = o.buf;
    m->bytes = o.bytes;
    HCK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
    HCK(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
    HCK(hipEventRecord(m->ready, o.stream));
    o.msg = m;
    World* w = o.comm->w;
    std::lock_guard<std::mutex> lk(w->mu);
    w->box[{o.comm->rank, o.peer}].push_back(m);
    w->n_send += 1;
    w->bytes += (long long)o.bytes;
    w->cv.notify_all();
  }
  // 2. every receive: wait for the matching send, copy on the receiver's stream
  for (Op& o : t_ops) {
    if (o.send) continue;
    World* w = o.comm->w;
    Msg* m = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      auto& q = w->box[{o.peer, o.comm->rank}];
      w->cv.wait(lk, [&] { return !q.empty(); });
      m = q.front();
      q.pop_front();
      w->n_recv += 1;
    }
    if (m->bytes != o.bytes) die("a matched send/recv pair disagrees on the count");
    HCK(hipStreamWaitEvent(o.stream, m->ready, 0));
    HCK(hipMemcpyAsync(o.buf, m->src, o.bytes, hipMemcpyDeviceToDevice, o.stream));
    HCK(hipEventRecord(m->done, o.stream));
    {
      std::lock_guard<std::mutex> lk(w->mu);
      m->consumed = true;
      w->cv.notify_all();
    }
  }
  // 3. a send is complete on the sender's stream once the receiver has copied
  for (Op& o : t_ops) {
    if (!o.send) continue;
    World* w = o.comm->w;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return o.msg->consumed; });
    }
    HCK(hipStreamWaitEvent(o.stream, o.msg->done, 0));
    // events are left to the process teardown: destroying one that another stream still waits on is not worth the risk here
    delete o.msg;
  }
  t_ops.clear();
  return 0;
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id, 0, sizeof(*id));
  const unsigned long long v = g_next_id++;
  memcpy(id->internal, &v, sizeof(v));
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  unsigned long long v;
  memcpy(&v, id.internal, sizeof(v));
  World* w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    World*& slot = g_worlds[v];
    if (!slot) {
      slot = new World();
      slot->nranks = nranks;
      slot->contrib.resize(nranks);
    }
    w = slot;
  }
  if (w->nranks != nranks || rank < 0 || rank >= nranks) return 4;
  Comm* c = new Comm{w, rank};
  *comm = c;
  std::unique_lock<std::mutex> lk(w->mu);              // collective, like the real call
  w->joined += 1;
  w->cv.notify_all();
  w->cv.wait(lk, [&] { return w->joined >= w->nranks; });
  return 0;
}
int ncclCommDestroy(void* comm) {
  delete static_cast<Comm*>(comm);
  return 0;
}
int ncclGroupStart() {
  t_depth += 1;
  return 0;
}
int ncclGroupEnd() {
  if (t_depth <= 0) return 5;
  t_depth -= 1;
  return t_depth == 0 ? flush() : 0;
}
static int post(bool send, void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  if (dtype != 8) die("only ncclDouble is expected");
  Comm* c = static_cast<Comm*>(comm);
  if (peer < 0 || peer >= c->w->nranks || peer == c->rank) die("bad peer");
  t_ops.push_back(Op{send, buf, count * sizeof(double), peer, c, stream, nullptr});
  return t_depth == 0 ? flush() : 0;
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(false, buf, count, dtype, peer, comm, stream);
}
int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  if (dtype != 8 || op != 0) die("only a sum of doubles is expected");
  Comm* c = static_cast<Comm*>(comm);
  World* w = c->w;
  std::vector<double> mine(count);
  HCK(hipStreamSynchronize(stream));
  HCK(hipMemcpy(mine.data(), sendbuf, count * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> result;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    const int gen = w->generation;
    w->contrib[c->rank] = mine;
    if (++w->arrived == w->nranks) {
      w->sum.assign(count, 0.0);
      for (int r = 0; r < w->nranks; ++r) {
        if (w->contrib[r].size() != count) die("all-reduce counts disagree between ranks");
        for (size_t i = 0; i < count; ++i) w->sum[i] += w->contrib[r][i];
      }
      w->arrived = 0;
      w->generation += 1;
      w->n_allreduce += 1;
      w->cv.notify_all();
    } else {
      w->cv.wait(lk, [&] { return w->generation != gen; });
    }
    result = w->sum;
  }
  HCK(hipMemcpy(recvbuf, result.data(), count * sizeof(double), hipMemcpyHostToDevice));
  // `sum` was read under the lock; the next round cannot complete (and overwrite it) before every rank has arrived again
  return 0;
}
const char* ncclGetErrorString(int code) {
  (void)code;
  return "mock_rccl error";
}
// what the test reads back: sends, receives, all-reduces, bytes sent over all ranks of every communicator
void mock_rccl_counters(long long* out4) {
  std::lock_guard<std::mutex> lk(g_mu);
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  for (auto& kv : g_worlds) {
    std::lock_guard<std::mutex> lk2(kv.second->mu);
    out4[0] += kv.second->n_send;
    out4[1] += kv.second->n_recv;
    out4[2] += kv.second->n_allreduce;
    out4[3] += kv.second->bytes;
  }
}
}
