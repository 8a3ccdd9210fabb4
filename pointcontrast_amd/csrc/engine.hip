// Network executor: a model lowered to a static program (tensors + ops) whose whole forward or
// backward is enqueued from C++ in one call.  See include/pcmi.h ("Network executor").
//
// Memory: one activation arena per pass (every tensor of the forward stays resident until that
// pass's backward: conv inputs, BN inputs and outputs are all needed again) and one gradient arena
// shared by the passes (backward passes run one after the other).  A tensor with parent >= 0 is a
// column slice of its parent's buffer, so a channel concatenation costs nothing: the producers
// write their slice (leading dimension = parent width) and the consumer reads the parent.
//
// Gradient accumulation: walking the ops in reverse, the first contribution to a (buffer, column
// range) overwrites, later ones accumulate inside the producing kernel's epilogue (no separate add
// kernels, no zero-fill of activation gradients).  Which is which is decided once, at creation.
#include <algorithm>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "internal.h"
#include "spconv_args.h"

namespace pcmi {

// PCMI_HOST_PROFILE=1: host-clock time per section of the executor calls, printed every 64 forwards (stderr)
struct HostProfile {
  bool on;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long calls = 0;
  std::chrono::steady_clock::time_point t;
  HostProfile() {
    const char* e = getenv("PCMI_HOST_PROFILE");
    on = e && e[0] == '1';
  }
  void start() {
    if (on) t = std::chrono::steady_clock::now();
  }
  void lap(int slot) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    acc[slot] += std::chrono::duration<double, std::milli>(now - t).count();
    t = now;
  }
  void report(const char* what, const char* const* names, int n) {
    if (!on || (++calls % 64) != 0) return;
    fprintf(stderr, "[pcmi host profile] %s, ms per call:", what);
    for (int i = 0; i < n; ++i) fprintf(stderr, " %s %.3f", names[i], acc[i] / (double)calls);
    fprintf(stderr, "\n");
  }
};
static HostProfile g_prof_fwd, g_prof_bwd;

// Host-side waits of the executor.  The wait in front of a forward pass for the previous pass's BatchNorm table is
// what keeps the enqueueing thread one pass ahead of the GPU, i.e. the thread sits in it for about half of every
// iteration.  Round 4 spun there (hipEventSynchronize: forward == forward_cpu == 8.9 ms of a 15.2 ms step in the
// bench line -- a core per rank burnt; with 8 ranks x (enqueue + draw + RCCL proxy threads) that is 8 cores next to
// the loader workers).  An event created with hipEventBlockingSync still spun on this driver (measured: same CPU time),
// so the wait polls the event and SLEEPS in between (PCMI_THROTTLE_SLEEP_US, default 50; 0 = hipEventSynchronize):
// the GPU is a whole backward pass behind the thread at that point, 50 us of wake-up latency cost nothing.
static unsigned host_wait_event_flags() {
  const char* e = getenv("PCMI_BLOCKING_EVENTS");  // 0: plain events (A/B)
  return hipEventDisableTiming | ((e && e[0] == '0') ? 0u : (unsigned)hipEventBlockingSync);
}
static int host_wait_event(hipEvent_t ev) {
  const char* e = getenv("PCMI_THROTTLE_SLEEP_US");
  const long us = e ? atol(e) : 50;
  if (us <= 0) {
    PCMI_HIP_CHECK(hipEventSynchronize(ev));
    return PCMI_OK;
  }
  // (tv_nsec must stay below one second: a larger PCMI_THROTTLE_SLEEP_US goes into tv_sec, or nanosleep returns EINVAL at once
  //  and the loop spins on hipEventQuery)
  const struct timespec ts = {(time_t)(us / 1000000L), (us % 1000000L) * 1000L};
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return PCMI_OK;
    if (q != hipErrorNotReady) PCMI_HIP_CHECK(q);
    // hipErrorNotReady is recorded as this thread's last error and PCMI_LAUNCH_CHECK must not see it -- but only THAT is
    // dropped: a genuine error of an earlier asynchronous launch stays for the next check to find
    if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
    nanosleep(&ts, nullptr);
  }
}

struct DevBuf {
  char* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;  // owns device memory
  DevBuf& operator=(const DevBuf&) = delete;
  // grows with 25 % head-room; the old block may still be in flight (on any of the executor's streams) -> sync
  // the device before freeing it (growth happens in the first iterations only)
  int reserve(size_t bytes, hipStream_t st) {
    if (bytes <= cap) return PCMI_OK;
    (void)st;
    if (p) {
      PCMI_HIP_CHECK(hipDeviceSynchronize());
      PCMI_HIP_CHECK(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    const size_t want = align_up(bytes + bytes / 4 + 4096, 1 << 20);
    PCMI_HIP_CHECK(hipMalloc((void**)&p, want));
    cap = want;
    return PCMI_OK;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

struct OpPlan {
  int acc_in = 0;   // gradient w.r.t. `in` accumulates (a previous op already wrote that range)
  int acc_res = 0;  // gradient w.r.t. `in2` (BN residual) accumulates
};

struct PassState {
  PassState() = default;
  PassState(const PassState&) = delete;  // owns device / pinned memory and events
  PassState& operator=(const PassState&) = delete;
  DevBuf act;
  DevBuf ws;  // forward workspace of this pass (passes may be forwarded concurrently on different streams)
  // deferred BatchNorm running-estimate updates of this pass (training & PCMI_NET_DEFER_RUNNING_STATS)
  BnRunningUpdate* upd_host = nullptr;  // pinned
  BnRunningUpdate* upd_dev = nullptr;
  int upd_cap = 0, upd_n = 0;
  hipEvent_t upd_copied = nullptr;
  // backward: gradient arena (activation gradients) and scratch of this pass
  DevBuf grad;
  std::vector<size_t> grad_off;
  DevBuf small;  // dgamma / dbeta scratch
  ~PassState() {
    for (int i = 0; i < 2; ++i) {
      if (x3_jobs_host[i]) (void)hipHostFree(x3_jobs_host[i]);
      if (x3_jobs_copied[i]) (void)hipEventDestroy(x3_jobs_copied[i]);
    }
    if (upd_host) (void)hipHostFree(upd_host);
    if (upd_dev) (void)hipFree(upd_dev);
    if (upd_copied) (void)hipEventDestroy(upd_copied);
    if (x3_bwd_packed) (void)hipEventDestroy(x3_bwd_packed);
  }
  std::vector<size_t> tensor_off;  // byte offset of every root tensor in `act`
  std::vector<size_t> stat_off;    // per op: BN save_mean/save_invstd (2*C floats) or L2 norms
  std::vector<size_t> bits_off;    // per op: the ReLU pattern of a BN(+residual)+ReLU output, one bit per element (norm.hip: relu_bits); SIZE_MAX = none
  std::vector<pcmi_kmap_t> maps;   // per op
  std::vector<char> has_map;
  std::vector<int64_t> rows;       // per level
  std::vector<int64_t> split;      // per level: rows of the first segment (= rows when the pass has one segment)
  const float* in_feats = nullptr;
  int64_t in_ld = 0;
  float* out_feats = nullptr;
  int64_t out_ld = 0;
  pcmi_coords_t* coords = nullptr;
  bool valid = false;
  // split-precision convolutions: the weights of every eligible layer, both orientations, packed by ONE launch at the
  // top of this pass's forward (x3_prepack) instead of one pack launch in front of every convolution.  Per PASS, not
  // per network: two passes of an iteration may fall into different size classes (different slice widths = different
  // pack layouts) and may be in flight on two streams at once; each reads its own packs and its own job table.
  DevBuf x3_packs, x3_jobs_dev;
  std::vector<X3Prepacked> x3_table;
  std::vector<int> x3_nts;  // slice width per (conv op, orientation) the table was built for
  const float* x3_params = nullptr;
  bool x3_current = false;  // the last forward of this pass packed (the packs are those of its weights)
  int x3_n_jobs = 0;
  int x3_first_op = -1;
  int64_t x3_items = 0;
  // The jobs are ordered forward orientations first: [0, x3_items_fwd) is packed at the top of the forward pass, on its
  // stream; the backward-data orientations [x3_items_fwd, x3_items) -- not needed before the backward pass -- are packed
  // on the executor's side stream while the forward pass is in its coarse levels (PCMI_X3_PACK_SPLIT=0: all at the top).
  int64_t x3_items_fwd = 0;
  bool x3_bwd_owed = false;          // this pass's forward still has to enqueue the backward orientations
  bool x3_bwd_pending = false;       // ... they are enqueued on the side stream and x3_bwd_packed is recorded behind them
  hipEvent_t x3_bwd_packed = nullptr;  // ... recorded behind them on the side stream: the backward pass waits for it
  // the job table goes up through pinned host memory with an asynchronous copy ON the pass's stream: ordered behind
  // the pack kernel of the previous table still queued there, and the enqueueing thread does not wait for the queue
  // to drain (a pageable hipMemcpy would).  Two host buffers: a rebuild waits only for the copy before the last.
  X3PackJob* x3_jobs_host[2] = {nullptr, nullptr};
  size_t x3_jobs_host_cap[2] = {0, 0};
  hipEvent_t x3_jobs_copied[2] = {nullptr, nullptr};
  int x3_jobs_slot = 0;
};

}  // namespace pcmi

struct pcmi_net {
  std::vector<pcmi_net_tensor_t> tensors;
  std::vector<pcmi_net_op_t> ops;
  std::vector<pcmi::OpPlan> plan;
  int input_tensor = -1, output_tensor = -1, n_levels = 0;
  std::vector<pcmi::PassState> passes;
  // side: the weight gradients of a backward pass (off the critical path: nothing downstream reads them) at the lowest
  // stream priority, with a workspace of their own
  // side[1] (PCMI_WGRAD_SIDE2=1, an experiment that lost: 254.4 against 260.2 pairs/s, profiles/r05a_*): the SMALL
  // weight-gradient launches (levels under 8192 rows, strided and 1x1 layers: ~45 latency-bound launches per step that use
  // a few dozen compute units each) on a stream of their own, beside the large level-1 launches instead of behind them
  static constexpr int kSides = 2;
  hipStream_t side[kSides] = {nullptr, nullptr};
  hipEvent_t ev_main[kSides] = {nullptr, nullptr}, ev_side[kSides] = {nullptr, nullptr};
  // what produced the gradient bucket the running `ready` callback is about: the chain up to the bucket's last op on
  // the backward stream, and the weight gradients enqueued so far on the side stream (pcmi_net_stream_wait_bucket)
  hipEvent_t ev_bkt_main = nullptr, ev_bkt_side[kSides] = {nullptr, nullptr};
  hipEvent_t ev_fwd_fork = nullptr, ev_fwd_join = nullptr, ev_fwd_pack = nullptr;  // forward pass: a block's residual branch on the side stream
  bool bkt_valid = false, bkt_side[kSides] = {false, false};
  pcmi::DevBuf ws_side[kSides];
  // weight gradients of the coarse levels collected for ONE launch per run of layers (spconv_wgrad.hip: wgrad_group_*)
  pcmi::WgradGroupBuilder* wgroup = nullptr;
  int64_t param_extent = 0;  // floats covered by the ops' parameters (rounded up to 4)
  // pcmi_net_time_ops: timing events around the convolution launches of selected ops INSIDE the passes (bench.py:
  // roofline.in_step_ms -- what the dominant kernel costs where it runs, next to the other streams, not stand-alone)
  // A ring of `timed_sets` event sets, one per forward pass enqueued since (its backward uses the same set): the
  // caller runs that many iterations WITHOUT synchronising and reads the sets afterwards.
  static constexpr int kTimedEv = 6;
  std::vector<int> timed_ops;
  std::vector<hipEvent_t> timed_ev;  // [set][timed op][forward begin / end, backward-data begin / end, weight-gradient begin / end (side stream)]
  std::vector<char> timed_hit;       // [set][timed op]: bit 0 forward, bit 1 backward-data, bit 2 weight gradient recorded
  int timed_sets = 0, timed_cur = -1;
  // pcmi_net_time_all: EVERY op is timed and the coarse-level weight gradients stay grouped (a timed op is otherwise launched
  // on its own); the grouped launches get events of their own: [set][flush][begin / end], timed_group_n[set] of them recorded
  static constexpr int kTimedGroups = 16;
  bool timed_keep_groups = false;
  std::vector<hipEvent_t> timed_group_ev;
  std::vector<int> timed_group_n;
  std::vector<int> timed_cnt;        // [set][timed op][forward / backward / weight gradient]: kernel launches of that call
  std::vector<int> timed_group_cnt;  // [set][flush]
  int timed_slot(int op) const {     // index into timed_hit (x kTimedEv: into timed_ev) of `op` in the current set, or -1
    if (timed_ops.empty() || timed_cur < 0) return -1;
    for (size_t q = 0; q < timed_ops.size(); ++q)
      if (timed_ops[q] == op) return (timed_cur % timed_sets) * (int)timed_ops.size() + (int)q;
    return -1;
  }
  void timed_clear() {
    for (hipEvent_t e : timed_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : timed_group_ev) (void)hipEventDestroy(e);
    timed_group_ev.clear();
    timed_group_n.clear();
    timed_cnt.clear();
    timed_group_cnt.clear();
    timed_keep_groups = false;
    timed_ev.clear();
    timed_ops.clear();
    timed_hit.clear();
    timed_sets = 0;
    timed_cur = -1;
  }
  ~pcmi_net() {
    (void)hipDeviceSynchronize();
    timed_clear();
    pcmi::wgrad_group_destroy(wgroup);
    for (int i = 0; i < kSides; ++i) {
      if (side[i]) (void)hipStreamDestroy(side[i]);
      if (ev_main[i]) (void)hipEventDestroy(ev_main[i]);
      if (ev_side[i]) (void)hipEventDestroy(ev_side[i]);
      if (ev_bkt_side[i]) (void)hipEventDestroy(ev_bkt_side[i]);
    }
    if (ev_bkt_main) (void)hipEventDestroy(ev_bkt_main);
    if (ev_fwd_fork) (void)hipEventDestroy(ev_fwd_fork);
    if (ev_fwd_join) (void)hipEventDestroy(ev_fwd_join);
    if (ev_fwd_pack) (void)hipEventDestroy(ev_fwd_pack);
  }
};

namespace pcmi {

static int root_of(const pcmi_net& n, int t) {
  while (n.tensors[t].parent >= 0) t = n.tensors[t].parent;
  return t;
}
static int col_of(const pcmi_net& n, int t) {
  int c = 0;
  while (n.tensors[t].parent >= 0) {
    c += n.tensors[t].col_off;
    t = n.tensors[t].parent;
  }
  return c;
}

struct View {
  float* p;
  int64_t ld;
};

static View act_view(const pcmi_net& n, const PassState& ps, int t) {
  if (t == n.input_tensor) return {const_cast<float*>(ps.in_feats), ps.in_ld};
  if (t == n.output_tensor) return {ps.out_feats, ps.out_ld};
  const int r = root_of(n, t);
  float* base = (float*)(ps.act.p + ps.tensor_off[r]);
  return {base + col_of(n, t), (int64_t)n.tensors[r].channels};
}

static View grad_view(const pcmi_net& n, const PassState& ps, int t, const float* d_out, int64_t d_ld) {
  if (t == n.output_tensor) return {const_cast<float*>(d_out), d_ld};
  const int r = root_of(n, t);
  float* base = (float*)(ps.grad.p + ps.grad_off[r]);
  return {base + col_of(n, t), (int64_t)n.tensors[r].channels};
}

static size_t op_workspace(const pcmi_net_op_t& op, int64_t n_in, int64_t n_out, int64_t M) {
  if (op.type == PCMI_OP_CONV) {
    const int K = op.kernel_size * op.kernel_size * op.kernel_size;
    return pcmi_spconv_workspace_bytes(n_in, n_out, op.cin, op.cout, K, M);
  }
  if (op.type == PCMI_OP_BN) return pcmi_bn_workspace_bytes(n_in, op.cout);
  return 0;
}

// Packs the weights of every layer the split-precision kernel can take (3^3 / 2^3 convolutions with >= 64 channels on
// both sides), forward and backward-data orientation, in one launch on `st`, and makes the table current for this
// thread's convolution calls.  The job table is rebuilt when the parameter buffer changes or a level crosses a size
// class (its slice width changes).  PCMI_X3_PREPACK=0: every convolution packs its own weights in front of its launch
// (as the C-ABI entry points do).
static int x3_prepack(pcmi_net& n, PassState& ps, const float* params, hipStream_t st, bool split_bwd) {
  const char* pe = getenv("PCMI_X3_PREPACK");  // read per pass: the parity test runs both forms in one process
  const bool enabled = !(pe && pe[0] == '0');
  ps.x3_current = false;
  ps.x3_bwd_pending = false;
  ps.x3_bwd_owed = false;
  if (!enabled || !pcmi_spconv_split_precision()) {
    x3_set_prepacked(nullptr, 0);
    return PCMI_OK;
  }
  const std::vector<int64_t>& rows = ps.rows;
  std::vector<int> nts;
  ps.x3_first_op = -1;  // the first op whose forward launch reads a pack (the forward pass joins a side-stream pack there)
  for (size_t oi = 0; oi < n.ops.size(); ++oi) {
    const auto& op = n.ops[oi];
    if (op.type != PCMI_OP_CONV || op.kernel_size <= 1) continue;
    const int K = op.kernel_size * op.kernel_size * op.kernel_size;
    for (int tr = 0; tr < 2; ++tr) {
      const int C = tr ? op.cout : op.cin, N = tr ? op.cin : op.cout;
      nts.push_back(x3_plan_nt(rows[n.tensors[tr ? op.in : op.out].level], C, N, K));
      if (tr == 0 && nts.back() >= 2 && ps.x3_first_op < 0) ps.x3_first_op = (int)oi;
    }
  }
  if (ps.x3_params != params || nts != ps.x3_nts) {
    size_t slot = 0;
    std::vector<X3PackJob> jobs;
    std::vector<X3Prepacked> table;
    size_t bytes = 0;
    int64_t items = 0;
    int64_t items_fwd = 0;
    for (int tr = 0; tr < 2; ++tr) {  // 0: B = W[k] ([cin x cout]); 1: B = W[k]^T ([cout x cin]) -- all forward jobs first
      slot = 0;
      if (tr == 1) items_fwd = items;
      for (const auto& op : n.ops) {
        if (op.type != PCMI_OP_CONV || op.kernel_size <= 1) continue;
        const int K = op.kernel_size * op.kernel_size * op.kernel_size;
        const int C = tr ? op.cout : op.cin, N = tr ? op.cin : op.cout;
        const int NT = nts[slot + tr];
        slot += 2;
        if (NT < 2) continue;
        X3PackJob j;
        j.w = params + op.w_off;
        j.w_kstride = (int64_t)op.cin * op.cout;
        j.w_sc = tr ? 1 : op.cout;
        j.w_sn = tr ? op.cout : 1;
        j.K = K;
        j.C = C;
        j.N = N;
        j.NS = 32 * NT;
        j.out = (void*)bytes;  // offset for now
        j.first_item = items;
        jobs.push_back(j);
        table.push_back({j.w, tr, NT, (const void*)bytes});
        bytes += x3_pack_bytes(K, C, N);
        items += (int64_t)K * (C / 32) * N * 4;
      }
    }
    ps.x3_items_fwd = items_fwd;
    if (!jobs.empty()) {
      // (a growing buffer drains the device first -- DevBuf::reserve -- so nothing in flight reads the old block)
      int rc = ps.x3_packs.reserve(bytes, st);
      if (rc) return rc;
      rc = ps.x3_jobs_dev.reserve(jobs.size() * sizeof(X3PackJob), st);
      if (rc) return rc;
      for (size_t i = 0; i < jobs.size(); ++i) {
        jobs[i].out = ps.x3_packs.p + (size_t)jobs[i].out;
        table[i].pack = ps.x3_packs.p + (size_t)table[i].pack;
      }
      const int hs = ps.x3_jobs_slot;
      ps.x3_jobs_slot ^= 1;
      const size_t need = jobs.size() * sizeof(X3PackJob);
      if (!ps.x3_jobs_copied[hs]) PCMI_HIP_CHECK(hipEventCreateWithFlags(&ps.x3_jobs_copied[hs], host_wait_event_flags()));
      PCMI_HIP_CHECK(hipEventSynchronize(ps.x3_jobs_copied[hs]));  // the copy that last read this host buffer (two tables ago)
      if (need > ps.x3_jobs_host_cap[hs]) {
        if (ps.x3_jobs_host[hs]) PCMI_HIP_CHECK(hipHostFree(ps.x3_jobs_host[hs]));
        ps.x3_jobs_host[hs] = nullptr;
        ps.x3_jobs_host_cap[hs] = 0;
        PCMI_HIP_CHECK(hipHostMalloc((void**)&ps.x3_jobs_host[hs], 2 * need, hipHostMallocDefault));
        ps.x3_jobs_host_cap[hs] = 2 * need;
      }
      memcpy(ps.x3_jobs_host[hs], jobs.data(), need);
      // in stream order: behind the previous table's pack kernel (and every convolution that read its packs) on `st`
      PCMI_HIP_CHECK(hipMemcpyAsync(ps.x3_jobs_dev.p, ps.x3_jobs_host[hs], need, hipMemcpyHostToDevice, st));
      PCMI_HIP_CHECK(hipEventRecord(ps.x3_jobs_copied[hs], st));
    }
    ps.x3_table.swap(table);
    ps.x3_nts.swap(nts);
    ps.x3_n_jobs = (int)jobs.size();
    ps.x3_items = items;
    ps.x3_params = params;
  }
  if (ps.x3_n_jobs == 0) {
    x3_set_prepacked(nullptr, 0);
    return PCMI_OK;
  }
  // (training passes only owe the backward orientations; `split`: they follow later on the side stream, x3_pack_backward)
  const int64_t upto = split_bwd ? ps.x3_items_fwd : ps.x3_items;
  const int rc = x3_pack_many(reinterpret_cast<const X3PackJob*>(ps.x3_jobs_dev.p), ps.x3_n_jobs, upto, st);
  if (rc) return rc;
  // (the forward pass still OWES the backward orientations; x3_bwd_pending -- "an event to wait for exists" -- is set only
  //  once they have been enqueued and the event recorded, pcmi_net_forward: a pass that fails in between must not leave a
  //  stale event to be waited for as if the packs were those of its weights -- ADVICE round 5)
  ps.x3_bwd_owed = split_bwd && ps.x3_items > ps.x3_items_fwd;
  ps.x3_current = true;
  x3_set_prepacked(ps.x3_table.data(), (int)ps.x3_table.size());
  return PCMI_OK;
}

// Clears the calling thread's table of packed weights on scope exit.  The backward form first makes the packs of the
// pass's last forward current again -- if that forward packed at all (x3_current): they are those of `params` as long
// as the weights have not been touched since (a backward pass differentiates the forward pass that produced them, so
// they have not).  A layer whose level changed its size class in between simply does not find its (weights,
// orientation, width) entry and packs for itself.
struct X3TableScope {
  X3TableScope() = default;
  X3TableScope(const PassState& ps, const float* params) {
    if (ps.x3_current && ps.x3_n_jobs > 0 && ps.x3_params == params) x3_set_prepacked(ps.x3_table.data(), (int)ps.x3_table.size());
  }
  X3TableScope(const X3TableScope&) = delete;
  X3TableScope& operator=(const X3TableScope&) = delete;
  ~X3TableScope() { x3_set_prepacked(nullptr, 0); }
};

// ---- backward -----------------------------------------------------------------------------------------
// One pass's backward is a chain (bwd-data -> BN-bwd -> ...) on the caller's stream plus the weight gradients, which
// nothing downstream reads, on a low-priority side stream with a workspace of their own.  (Round 2 also carried a
// two-chain form that ran the backward passes of the two clouds next to each other -- bit-identical, 111 against 174
// pairs/s, a queueing effect of five concurrently active streams; the pair as ONE two-segment pass replaced it and the
// code is gone.)
struct BackwardJob {
  int pass = 0;
  const float* d_out = nullptr;
  int64_t d_ld = 0;
  hipStream_t st = nullptr;
};

// Test hooks of the bucket hand-over (tests/test_gpu_bucket_sync.py; read per call, never set in production):
//   PCMI_DEBUG_SIDE_DELAY_US=<n>        the weight-gradient stream starts every backward pass n microseconds late (a spin
//                                       kernel at its head), so that it trails the chain by far more than it ever does
//   PCMI_DEBUG_SKIP_BUCKET_SIDE_WAIT=1  pcmi_net_stream_wait_bucket leaves out the wait for the weight-gradient stream --
//                                       the negative control: with the delay above a consumer must then see stale gradients
__global__ void debug_delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();  // constant-rate counter (100 MHz)
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static long debug_env_long(const char* name) {
  const char* e = getenv(name);
  return e ? atol(e) : 0;
}

static int ensure_streams(pcmi_net& n, bool second) {
  if (n.side[0] && (!second || n.side[1])) return PCMI_OK;
  // the weight gradients are off the critical path: lowest priority, so that the chain's kernels are dispatched
  // first and the weight-gradient workgroups fill what they leave idle (same priority measured: 17.5-19.4 against
  // 16.6 ms per iteration; weight gradients on the chain's own stream: 28 ms)
  int least = 0, greatest = 0;
  PCMI_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  // (a stream confined to a subset of the compute units -- hipExtStreamCreateWithCUMask, so that the chain's small
  //  kernels always find free units -- was tried: such a stream cannot be non-blocking and serialises against the
  //  caller's default stream, 165 against 253 pairs/s with any mask; profiles/r03j_bench_ab_cu_mask.txt)
  // The second side stream exists ONLY under PCMI_WGRAD_SIDE2=1.  Round 5 created it unconditionally for one GPU call
  // (idle unless the switch was on) and the forced 1-rank reducer run went from 15.8 to 45 ms per step: with the
  // reducer's communication stream it was one stream too many for the device's hardware queues, two of the executor's
  // streams shared a queue and serialised (profiles/r05b_idle_second_side_stream_and_forced_reducer.txt).  Streams of
  // a rank in the step: compute, plan, weight-gradient side, communication -- nothing else may be added casually.
  for (int i = 0; i < (second ? pcmi_net::kSides : 1); ++i) {
    if (n.side[i]) continue;
    PCMI_HIP_CHECK(hipStreamCreateWithPriority(&n.side[i], hipStreamNonBlocking, least));
    PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_main[i], hipEventDisableTiming));
    PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_side[i], hipEventDisableTiming));
    PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_bkt_side[i], hipEventDisableTiming));
  }
  if (!n.ev_bkt_main) PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_bkt_main, hipEventDisableTiming));
  return PCMI_OK;
}

// One pass's backward: op by op in reverse, the chain on `st`, the weight gradients on the side stream.
struct BackwardRun {
  pcmi_net& n;
  BackwardJob job;
  const float* params;
  float* grads;
  const int64_t* bucket_lo_host;
  int n_buckets;
  pcmi_ready_fn ready;
  void* ready_ctx;
  // set by begin()
  hipStream_t st = nullptr, wst = nullptr;  // wst: the side stream of the op being differentiated (step)
  bool two_sides = false;
  bool stem_on_chain = true;  // the input layer's weight gradient on the chain (see step())
  int64_t small_rows = 0;
  PassState* ps = nullptr;
  DevBuf* wws = nullptr;
  bool pending[pcmi_net::kSides] = {false, false}, used[pcmi_net::kSides] = {false, false};
  float* scratch_g = nullptr;
  size_t scratch_stride = 0;
  int bn_seen = 0;
  bool bn_par = false;
  std::vector<int> bucket_last;

  BackwardRun(pcmi_net& net, const BackwardJob& j, const float* prm, float* g, const int64_t* blo, int nb, pcmi_ready_fn r,
              void* rc)
      : n(net), job(j), params(prm), grads(g), bucket_lo_host(blo), n_buckets(nb), ready(r), ready_ctx(rc) {}

  int join_side() {  // `st` continues only after the weight gradients enqueued so far
    for (int q = 0; q < pcmi_net::kSides; ++q) {
      if (!pending[q]) continue;
      PCMI_HIP_CHECK(hipEventRecord(n.ev_side[q], n.side[q]));
      PCMI_HIP_CHECK(hipStreamWaitEvent(st, n.ev_side[q], 0));
      pending[q] = false;
    }
    return PCMI_OK;
  }
  // The collected weight gradients go to the side stream now: everything their operands need has been enqueued on `st`.
  int flush_group() {
    if (wgrad_group_size(n.wgroup) == 0) return PCMI_OK;
    PCMI_HIP_CHECK(hipEventRecord(n.ev_main[0], st));
    PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_main[0], 0));
    pending[0] = used[0] = true;
    int slot = -1;  // (pcmi_net_time_all: the grouped launch between two events of the current set)
    if (n.timed_keep_groups && n.timed_cur >= 0) {
      const int set = n.timed_cur % n.timed_sets;
      if (n.timed_group_n[set] < pcmi_net::kTimedGroups) slot = set * pcmi_net::kTimedGroups + n.timed_group_n[set];
    }
    if (slot >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_group_ev[2 * slot], n.side[0]));
    const long l0 = g_launches;
    const int rc = wgrad_group_flush(n.wgroup, n.side[0]);
    if (slot >= 0 && !rc) {
      n.timed_group_cnt[slot] = (int)(g_launches - l0);
      PCMI_HIP_CHECK(hipEventRecord(n.timed_group_ev[2 * slot + 1], n.side[0]));
      ++n.timed_group_n[n.timed_cur % n.timed_sets];
    }
    return rc;
  }
  int bucket_of(int64_t offp) const {
    int b = 0;
    for (int q = 0; q < n_buckets; ++q)
      if (offp >= bucket_lo_host[q]) b = q;
    return b;
  }

  int begin() {
    st = job.st;
    ps = &n.passes[job.pass];
    const int n_ops = (int)n.ops.size(), n_t = (int)n.tensors.size();
    if (n_buckets < 0 || !bucket_lo_host) n_buckets = 0;
    ps->grad_off.assign(n_t, 0);
    size_t off = 0;
    int max_c = 4;
    for (int t = 0; t < n_t; ++t) {
      max_c = std::max(max_c, n.tensors[t].channels);
      if (n.tensors[t].parent >= 0 || t == n.input_tensor || t == n.output_tensor) continue;
      ps->grad_off[t] = off;
      off += align_up((size_t)ps->rows[n.tensors[t].level] * n.tensors[t].channels * sizeof(float), 256);
    }
    int rc = ps->grad.reserve(off, st);
    if (rc) return rc;
    // per BatchNorm op its own [2 segments][dbeta, dgamma] sums: a BatchNorm whose parameter gradients are accumulated on
    // the side stream (bn_param_accumulate) must not have its sums overwritten by the next BatchNorm of the chain
    int n_bn = 0;
    for (int i = 0; i < n_ops; ++i) n_bn += n.ops[i].type == PCMI_OP_BN;
    rc = ps->small.reserve((size_t)(n_bn + 1) * 4 * max_c * sizeof(float) + 256, st);
    if (rc) return rc;
    scratch_g = (float*)ps->small.p;
    scratch_stride = 4 * (size_t)max_c;
    bn_seen = 0;
    {  // PCMI_BN_SMALL_PAR=1: a small BatchNorm's backward with its two segments side by side and the parameter gradients
       // added by a kernel on the side stream.  Measured (profiles/r05h_*): 272.5 against 275.9 pairs/s with the segments one
       // after the other inside the kernel -- 30 more launches and event pairs on the weight-gradient stream cost more than
       // the shorter BatchNorm kernels return.  Off.
      const char* e = getenv("PCMI_BN_SMALL_PAR");
      bn_par = e && e[0] == '1';
    }
    two_sides = debug_env_long("PCMI_WGRAD_SIDE2") != 0;
    {
      const char* e = getenv("PCMI_STEM_WGRAD_ON_CHAIN");
      stem_on_chain = !(e && e[0] == '0');
    }
    rc = ensure_streams(n, two_sides);
    if (rc) return rc;
    small_rows = 8192;  // (= the default of PCMI_WGRAD_X3T: what is under it takes the pair-list kernel)
    for (int q = 0; q < (two_sides ? 2 : 1); ++q) {
      rc = n.ws_side[q].reserve(ps->ws.cap, n.side[q]);
      if (rc) return rc;
      pending[q] = used[q] = false;
    }
    wst = n.side[0];
    wws = &n.ws_side[0];
    if (!n.wgroup) n.wgroup = wgrad_group_create();
    wgrad_group_drop(n.wgroup);
    if (ps->x3_bwd_pending) {  // the backward-data weight packs of this pass were enqueued on the side stream (forward)
      PCMI_HIP_CHECK(hipStreamWaitEvent(st, ps->x3_bwd_packed, 0));
      ps->x3_bwd_pending = false;
    }
    if (const long us = debug_env_long("PCMI_DEBUG_SIDE_DELAY_US"); us > 0) {
      debug_delay_kernel<<<1, 1, 0, wst>>>((long long)us * 100);
      PCMI_LAUNCH_CHECK();
    }
    n.bkt_valid = false;
    // bucket -> first op (lowest index) that owns parameters of it: the bucket is final after that op
    bucket_last.assign(n_buckets, -1);
    for (int i = n_ops - 1; i >= 0 && n_buckets > 0; --i) {
      const auto& op = n.ops[i];
      if (op.type == PCMI_OP_L2NORM) continue;
      bucket_last[bucket_of(op.w_off)] = i;
      if (op.type == PCMI_OP_BN || op.has_bias) bucket_last[bucket_of(op.b_off)] = i;
    }
    return PCMI_OK;
  }

  int step(int i) {  // differentiate op i (called for i = n_ops - 1 ... 0)
    const float* d_out = job.d_out;
    const int64_t d_ld = job.d_ld;
    const auto& op = n.ops[i];
    const OpPlan& pl = n.plan[i];
    const View x = act_view(n, *ps, op.in), y = act_view(n, *ps, op.out);
    const View dy = grad_view(n, *ps, op.out, d_out, d_ld);
    const int64_t n_in = ps->rows[n.tensors[op.in].level], n_out = ps->rows[n.tensors[op.out].level];
    int rc = PCMI_OK;
    if (op.type == PCMI_OP_CONV) {
      const pcmi_kmap_t* map = ps->has_map[i] ? &ps->maps[i] : nullptr;
      // dy is complete at this point of `st` (all its consumers were differentiated before)
      // (weight gradients of the largest layers ON the chain instead of the side stream -- they cannot share a CU with the
      //  chain's kernels anyway, two of their workgroups fill register file and LDS -- were measured: 16.47 against 15.85 ms
      //  per step with the 175k-row layers in the chain, 17.1 with everything from 8000 rows: profiles/r04i_*)
      // a coarse-level 3^3 layer: its gradient joins the open group and is launched with the others of its run (at the
      // next layer that does not fit the group, at a bucket boundary, or at the end of the pass)
      bool grouped = false;
      // PCMI_DEBUG_SKIP_WGRAD_ROWS=<n> (timing diagnostic, WRONG gradients): the weight gradients of layers with at least n
      // rows on both sides are not launched at all (1 = none is) -- what the chain costs without them beside it
      // (profiles/r06a_*: the bound of moving the level-1 weight gradients out of the backward pass)
      const long skip_rows = debug_env_long("PCMI_DEBUG_SKIP_WGRAD_ROWS");
      const bool skip_wgrad = skip_rows > 0 && std::min(n_in, n_out) >= skip_rows && op.in != n.input_tensor;
      if (!skip_wgrad && !two_sides && !op.transpose && op.kernel_size == 3 && op.stride == 1 && !op.has_bias && map && (n.timed_keep_groups || n.timed_slot(i) < 0)) {
        grouped = wgrad_group_add(n.wgroup, x.p, x.ld, n_in, op.cin, dy.p, dy.ld, n_out, op.cout, map, grads + op.w_off, 1, n.side[0]);
        if (!grouped && wgrad_group_size(n.wgroup) > 0) {  // another tile shape, or the group is full: launch it, start a new one
          rc = flush_group();
          if (rc) return rc;
          grouped = wgrad_group_add(n.wgroup, x.p, x.ld, n_in, op.cin, dy.p, dy.ld, n_out, op.cout, map, grads + op.w_off, 1, n.side[0]);
        }
      }
      if (!grouped && !skip_wgrad) {
      const int sq = (two_sides && (std::min(n_in, n_out) < small_rows || op.kernel_size != 3 || op.stride != 1 ||
                                    std::min(op.cin, op.cout) < 64)) ? 1 : 0;
      // The layer that reads the network's input is the LAST op of a backward pass: nothing is left on the chain that its
      // weight gradient could run beside, the pass's join follows.  On the side stream it cost two event hand-overs
      // (chain -> side -> chain, ~10 us each in the kernel trace) for 50 us of work: it stays on the chain, with the chain's
      // workspace.  PCMI_STEM_WGRAD_ON_CHAIN=0: on the side stream like every other weight gradient (A/B; read per pass).
      const bool on_chain = op.in == n.input_tensor && stem_on_chain;
      if (on_chain) {
        wst = st;
        wws = &ps->ws;
      } else {
        wst = n.side[sq];
        wws = &n.ws_side[sq];
        PCMI_HIP_CHECK(hipEventRecord(n.ev_main[sq], st));
        PCMI_HIP_CHECK(hipStreamWaitEvent(wst, n.ev_main[sq], 0));
        pending[sq] = used[sq] = true;
      }
      const int tw = n.timed_slot(i);
      if (tw >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tw + 4], wst));
      const long lw0 = g_launches;
      rc = spconv_backward_weight(x.p, x.ld, n_in, op.cin, dy.p, dy.ld, n_out, op.cout, map, op.transpose, grads + op.w_off,
                                  op.has_bias ? grads + op.b_off : nullptr, 1, wws->p, wws->cap, wst);
      if (rc) return rc;
      if (tw >= 0) {
        PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tw + 5], wst));
        n.timed_hit[tw] |= 4;
        n.timed_cnt[3 * tw + 2] = (int)(g_launches - lw0);
      }
      }  // !grouped
      if (op.in != n.input_tensor) {
        const View dx = grad_view(n, *ps, op.in, d_out, d_ld);
        const int tq = n.timed_slot(i);
        if (tq >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tq + 2], st));
        const long lb0 = g_launches;
        rc = spconv_backward_data(dy.p, dy.ld, n_out, op.cout, params + op.w_off, op.cin, map, op.transpose, dx.p, dx.ld,
                                  n_in, pl.acc_in, ps->ws.p, ps->ws.cap, st);
        if (tq >= 0 && !rc) {
          PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tq + 3], st));
          n.timed_hit[tq] |= 2;
          n.timed_cnt[3 * tq + 1] = (int)(g_launches - lb0);
        }
      }
    } else if (op.type == PCMI_OP_BN) {
      const int tb = n.timed_slot(i);
      if (tb >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tb + 2], st));
      const long lb0 = g_launches;
      const View dx = grad_view(n, *ps, op.in, d_out, d_ld);
      View dr = {nullptr, 0};
      if (op.in2 >= 0) dr = grad_view(n, *ps, op.in2, d_out, d_ld);
      const float* stats0 = (const float*)(ps->act.p + ps->stat_off[i]);
      const uint32_t* rbits = (op.relu && ps->bits_off[i] != SIZE_MAX) ? (const uint32_t*)(ps->act.p + ps->bits_off[i]) : nullptr;
      const int64_t sp = ps->split[n.tensors[op.in].level];
      if (sp < n_in) {  // a segment = one forward call of the reference: own statistics, own sums; one launch pair
        float* sums = scratch_g + (size_t)(++bn_seen) * scratch_stride;
        int deferred = 0;
        rc = bn_backward2(dy.p, dy.ld, x.p, x.ld, op.relu ? y.p : nullptr, y.ld, n_in, sp, op.cout, params + op.w_off, stats0,
                          stats0 + op.cout, 3 * op.cout, dx.p, dx.ld, dr.p, dr.ld, pl.acc_res, sums, grads + op.w_off,
                          grads + op.b_off, ps->ws.p, ps->ws.cap, st, (two_sides || !bn_par) ? nullptr : &deferred, rbits);
        if (!rc && deferred) {  // the parameter gradients of this BatchNorm: on the side stream, behind the sums
          PCMI_HIP_CHECK(hipEventRecord(n.ev_main[0], st));
          PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_main[0], 0));
          pending[0] = used[0] = true;
          rc = bn_param_accumulate(sums, op.cout, grads + op.w_off, grads + op.b_off, n.side[0]);
        }
      } else
        rc = bn_backward(dy.p, dy.ld, x.p, x.ld, op.relu ? y.p : nullptr, y.ld, n_in, op.cout, params + op.w_off, stats0,
                         stats0 + op.cout, dx.p, dx.ld, dr.p, dr.ld, pl.acc_res, scratch_g, scratch_g + op.cout,
                         grads + op.w_off, grads + op.b_off, ps->ws.p, ps->ws.cap, st, rbits);
      if (tb >= 0 && !rc) {
        PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tb + 3], st));
        n.timed_hit[tb] |= 2;
        n.timed_cnt[3 * tb + 1] = (int)(g_launches - lb0);
      }
    } else {
      const View dx = grad_view(n, *ps, op.in, d_out, d_ld);
      const int tl = n.timed_slot(i);
      if (tl >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tl + 2], st));
      const long ll0 = g_launches;
      rc = pcmi_l2norm_bwd(dy.p, dy.ld, y.p, y.ld, (const float*)(ps->act.p + ps->stat_off[i]), n_in, op.cout, dx.p, dx.ld,
                           (pcmi_stream_t)st);
      if (tl >= 0 && !rc) {
        PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tl + 3], st));
        n.timed_hit[tl] |= 2;
        n.timed_cnt[3 * tl + 1] = (int)(g_launches - ll0);
      }
    }
    if (rc) return rc;
    // A bucket is final once this op's kernels have run -- on `st` AND, for the weight gradients, on the side stream.
    // The chain does NOT wait for the side stream here (round 3 joined the two at every bucket boundary: with five
    // buckets the backward chain stalled five times per step behind a weight-gradient stream that runs behind it by
    // design -- and only when a reducer was attached, so the 1-GPU step did not predict the per-GPU step at N > 1).
    // Both positions are recorded as events instead and the consumer orders ITS stream behind them from inside the
    // callback (pcmi_net_stream_wait_bucket); the one join of the pass stays in end(), in front of the optimiser.
    for (int b = 0; b < n_buckets; ++b)
      if (bucket_last[b] == i && ready) {
        rc = flush_group();  // (collected gradients may belong to this bucket: they must be on the side stream before its event)
        if (rc) return rc;
        PCMI_HIP_CHECK(hipEventRecord(n.ev_bkt_main, st));
        for (int q = 0; q < pcmi_net::kSides; ++q) {
          if (used[q]) PCMI_HIP_CHECK(hipEventRecord(n.ev_bkt_side[q], n.side[q]));
          n.bkt_side[q] = used[q];
        }
        n.bkt_valid = true;
        ready(ready_ctx, b);
      }
    return PCMI_OK;
  }

  int end() {
    n.bkt_valid = false;
    int rc = flush_group();
    if (rc) return rc;
    rc = join_side();
    if (rc) return rc;
    ps->valid = false;
    return PCMI_OK;
  }
};

static int run_backward(pcmi_net& n, const BackwardJob& job, const float* params, float* grads,
                        const int64_t* bucket_lo_host, int n_buckets, pcmi_ready_fn ready, void* ready_ctx) {
  const X3TableScope x3_scope(n.passes[job.pass], params);
  BackwardRun r(n, job, params, grads, bucket_lo_host, n_buckets, ready, ready_ctx);
  int rc = r.begin();
  for (int i = (int)n.ops.size() - 1; i >= 0 && !rc; --i) rc = r.step(i);
  return rc ? rc : r.end();
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

int pcmi_net_create(const pcmi_net_tensor_t* tensors, int n_tensors, const pcmi_net_op_t* ops, int n_ops,
                    int input_tensor, int output_tensor, int n_passes, pcmi_net_t** out) {
  PCMI_REQUIRE(tensors && ops && out && n_tensors > 0 && n_ops > 0 && n_passes > 0 && n_passes <= 8, PCMI_ERR_INVALID,
               "net_create: bad argument");
  PCMI_REQUIRE(input_tensor >= 0 && input_tensor < n_tensors && output_tensor >= 0 && output_tensor < n_tensors,
               PCMI_ERR_INVALID, "net_create: bad input/output tensor id");
  pcmi_net* n = new pcmi_net();
  n->tensors.assign(tensors, tensors + n_tensors);
  n->ops.assign(ops, ops + n_ops);
  n->plan.resize(n_ops);
  n->input_tensor = input_tensor;
  n->output_tensor = output_tensor;
  n->passes = std::vector<pcmi::PassState>(n_passes);  // elements are neither copied nor moved afterwards
  auto fail = [&](const char* msg, int i) {
    set_error("net_create: %s (index %d)", msg, i);
    delete n;
    return PCMI_ERR_INVALID;
  };
  for (int t = 0; t < n_tensors; ++t) {
    const auto& T = n->tensors[t];
    if (T.level < 0 || T.level > 12 || T.channels <= 0) return fail("bad tensor", t);
    if (T.parent >= 0) {
      if (T.parent >= n_tensors || T.parent == t) return fail("bad parent", t);
      const auto& P = n->tensors[T.parent];
      if (P.level != T.level || T.col_off < 0 || T.col_off + T.channels > P.channels || T.col_off % 4 != 0)
        return fail("slice does not fit its parent (or is not 16-byte aligned)", t);
    }
    n->n_levels = std::max(n->n_levels, T.level + 1);
  }
  if (n->tensors[input_tensor].parent >= 0 || n->tensors[output_tensor].parent >= 0)
    return fail("the network input / output must be root tensors", input_tensor);
  // decide overwrite vs accumulate for every gradient contribution, walking backward
  std::vector<std::vector<char>> touched(n_tensors);
  for (int t = 0; t < n_tensors; ++t)
    if (n->tensors[t].parent < 0) touched[t].assign(n->tensors[t].channels, 0);
  auto contribute = [&](int t, int* acc_flag) -> bool {
    const int r = root_of(*n, t), c0 = col_of(*n, t), c1 = c0 + n->tensors[t].channels;
    int cnt = 0;
    for (int c = c0; c < c1; ++c) cnt += touched[r][c];
    if (cnt != 0 && cnt != c1 - c0) return false;  // partially written range: not expressible
    *acc_flag = cnt != 0;
    for (int c = c0; c < c1; ++c) touched[r][c] = 1;
    return true;
  };
  for (int i = n_ops - 1; i >= 0; --i) {
    const auto& op = n->ops[i];
    if (op.in < 0 || op.in >= n_tensors || op.out < 0 || op.out >= n_tensors || op.in2 >= n_tensors)
      return fail("bad tensor id in op", i);
    if (op.type == PCMI_OP_CONV) {
      const int64_t K = (int64_t)op.kernel_size * op.kernel_size * op.kernel_size;
      n->param_extent = std::max<int64_t>(n->param_extent, op.w_off + K * op.cin * op.cout);
      if (op.has_bias) n->param_extent = std::max<int64_t>(n->param_extent, op.b_off + op.cout);
    } else if (op.type == PCMI_OP_BN) {
      n->param_extent = std::max<int64_t>(n->param_extent, std::max(op.w_off, op.b_off) + op.cout);
    }
    if (op.type == PCMI_OP_CONV) {
      if (n->tensors[op.in].channels != op.cin || n->tensors[op.out].channels != op.cout) return fail("conv channels", i);
      if (op.in != input_tensor && !contribute(op.in, &n->plan[i].acc_in)) return fail("mixed gradient coverage", i);
    } else if (op.type == PCMI_OP_BN) {
      if (n->tensors[op.in].channels != op.cout || n->tensors[op.out].channels != op.cout) return fail("bn channels", i);
      if (op.in2 >= 0 && !contribute(op.in2, &n->plan[i].acc_res)) return fail("mixed gradient coverage", i);
      if (!contribute(op.in, &n->plan[i].acc_in) || n->plan[i].acc_in) return fail("BN input must have a single consumer", i);
    } else if (op.type == PCMI_OP_L2NORM) {
      if (!contribute(op.in, &n->plan[i].acc_in) || n->plan[i].acc_in) return fail("L2NORM input must have a single consumer", i);
    } else {
      return fail("unknown op type", i);
    }
  }
  n->param_extent = (n->param_extent + 3) / 4 * 4;
  *out = n;
  return PCMI_OK;
}

int pcmi_net_destroy(pcmi_net_t* net) {
  delete net;
  return PCMI_OK;
}

int pcmi_net_export_tensor(pcmi_net_t* net, int pass, int tensor, int64_t* rows, int* channels, float* out, int64_t out_ld,
                           pcmi_stream_t stream) {
  PCMI_REQUIRE(net && pass >= 0 && pass < (int)net->passes.size() && tensor >= 0 && tensor < (int)net->tensors.size(),
               PCMI_ERR_INVALID, "net_export_tensor: bad pass / tensor id");
  const PassState& ps = net->passes[pass];
  const auto& T = net->tensors[tensor];
  PCMI_REQUIRE(!ps.rows.empty() && ps.act.p, PCMI_ERR_INVALID, "net_export_tensor: pass %d has not been forwarded", pass);
  const int64_t n = ps.rows[T.level];
  if (rows) *rows = n;
  if (channels) *channels = T.channels;
  if (!out || n == 0) return PCMI_OK;
  PCMI_REQUIRE(out_ld >= T.channels, PCMI_ERR_INVALID, "net_export_tensor: out_ld %lld < %d channels", (long long)out_ld, T.channels);
  const View v = act_view(*net, ps, tensor);
  PCMI_REQUIRE(v.p, PCMI_ERR_INVALID, "net_export_tensor: tensor %d has no storage in this pass", tensor);
  PCMI_HIP_CHECK(hipMemcpy2DAsync(out, sizeof(float) * out_ld, v.p, sizeof(float) * v.ld, sizeof(float) * T.channels, (size_t)n,
                                  hipMemcpyDeviceToDevice, as_stream(stream)));
  return PCMI_OK;
}

int pcmi_net_memory_bytes(pcmi_net_t* net, size_t* bytes) {
  PCMI_REQUIRE(net && bytes, PCMI_ERR_INVALID, "net_memory_bytes: null argument");
  size_t b = net->ws_side[0].cap + net->ws_side[1].cap;
  for (auto& p : net->passes) b += p.act.cap + p.ws.cap + p.grad.cap + p.small.cap + p.x3_packs.cap + p.x3_jobs_dev.cap;
  *bytes = b;
  return PCMI_OK;
}

int pcmi_net_forward(pcmi_net_t* net, int pass, pcmi_coords_t* coords, const float* in_feats, int64_t in_ld,
                     int64_t n_rows, const float* params, int training, float* out_feats, int64_t out_ld,
                     pcmi_stream_t stream) {
  PCMI_REQUIRE(net && coords && in_feats && params && out_feats, PCMI_ERR_INVALID, "net_forward: null argument");
  PCMI_REQUIRE(pass >= 0 && pass < (int)net->passes.size(), PCMI_ERR_INVALID, "net_forward: bad pass %d", pass);
  hipStream_t st = as_stream(stream);
  pcmi_net& n = *net;
  PassState& ps = n.passes[pass];
  ps.valid = false;
  const int n_ops = (int)n.ops.size(), n_t = (int)n.tensors.size();
  g_prof_fwd.start();
  // the coordinate plan may have been enqueued on another stream without a host synchronisation behind it
  int rc = coords_wait_plan(coords, st);
  if (rc) return rc;
  // ---- level sizes (cache hits once the coordinate plan exists) ------------------------------------
  std::vector<int> keys(n.n_levels, 0);
  ps.rows.assign(n.n_levels, 0);
  int64_t n0 = 0;
  rc = pcmi_coords_size(coords, 0, &n0, nullptr);
  if (rc) return rc;
  PCMI_REQUIRE(n0 == n_rows, PCMI_ERR_INVALID, "net_forward: %lld feature rows but %lld coordinates", (long long)n_rows,
               (long long)n0);
  ps.rows[0] = n0;
  for (int l = 1; l < n.n_levels; ++l) {
    rc = pcmi_coords_stride(coords, keys[l - 1], 2, &keys[l], &ps.rows[l], stream);
    if (rc) return rc;
  }
  // two-segment batch (pcmi_coords_set_split): BatchNorm normalises every segment with its own statistics, as two
  // forward calls of the reference would
  ps.split.assign(n.n_levels, 0);
  for (int l = 0; l < n.n_levels; ++l) {
    int64_t sp = -1;
    rc = pcmi_coords_split(coords, keys[l], &sp);
    if (rc) return rc;
    ps.split[l] = (sp <= 0 || sp >= ps.rows[l]) ? ps.rows[l] : sp;
  }
  g_prof_fwd.lap(0);
  // ---- maps, arena layout, workspace --------------------------------------------------------------
  ps.maps.resize(n_ops);
  ps.has_map.assign(n_ops, 0);
  size_t ws_need = 0;
  for (int i = 0; i < n_ops; ++i) {
    const auto& op = n.ops[i];
    const int li = n.tensors[op.in].level, lo = n.tensors[op.out].level;
    int64_t M = ps.rows[li];
    if (op.type == PCMI_OP_CONV && op.kernel_size > 1) {
      if (!op.transpose && op.stride == 1) {
        PCMI_REQUIRE(li == lo, PCMI_ERR_INVALID, "net: op %d: stride-1 conv across levels", i);
        rc = kmap_get_nosync(coords, keys[li], keys[li], op.kernel_size, 1, op.region, &ps.maps[i], stream);
      } else if (!op.transpose) {
        PCMI_REQUIRE(lo == li + 1, PCMI_ERR_INVALID, "net: op %d: strided conv must go one level down", i);
        rc = kmap_get_nosync(coords, keys[li], keys[lo], op.kernel_size, op.stride, op.region, &ps.maps[i], stream);
      } else {
        PCMI_REQUIRE(lo == li - 1, PCMI_ERR_INVALID, "net: op %d: transposed conv must go one level up", i);
        rc = kmap_get_nosync(coords, keys[lo], keys[li], op.kernel_size, op.stride, op.region, &ps.maps[i], stream);
      }
      if (rc) return rc;
      ps.has_map[i] = 1;
      M = kmap_pairs_bound(ps.maps[i]);
    }
    ws_need = std::max(ws_need, op_workspace(op, ps.rows[li], ps.rows[lo], M));
  }
  g_prof_fwd.lap(1);
  ps.tensor_off.assign(n_t, 0);
  size_t off = 0;
  for (int t = 0; t < n_t; ++t) {
    if (n.tensors[t].parent >= 0 || t == n.input_tensor || t == n.output_tensor) continue;
    ps.tensor_off[t] = off;
    off += align_up((size_t)ps.rows[n.tensors[t].level] * n.tensors[t].channels * sizeof(float), 256);
  }
  ps.stat_off.assign(n_ops, 0);
  ps.bits_off.assign(n_ops, SIZE_MAX);
  // PCMI_BN_RELU_BITS=0: the backward BatchNorm kernels read the ReLU pattern from the fp32 output again (A/B; read per pass)
  const bool relu_bits_on = [] {
    const char* e = getenv("PCMI_BN_RELU_BITS");
    return !(e && e[0] == '0');
  }();
  for (int i = 0; i < n_ops; ++i) {
    const auto& op = n.ops[i];
    ps.stat_off[i] = off;
    if (op.type == PCMI_OP_BN) off += align_up((size_t)2 * 3 * op.cout * sizeof(float), 256);  // per segment: mean, invstd, unbiased var
    if (op.type == PCMI_OP_L2NORM) off += align_up((size_t)ps.rows[n.tensors[op.in].level] * sizeof(float), 256);
    if (op.type == PCMI_OP_BN && op.relu && op.cout % 32 == 0 && relu_bits_on && (training & 1)) {
      ps.bits_off[i] = off;
      off += align_up((size_t)ps.rows[n.tensors[op.in].level] * (op.cout / 32) * sizeof(uint32_t), 256);
    }
  }
  rc = ps.act.reserve(off, st);
  if (rc) return rc;
  rc = ps.ws.reserve(ws_need + 256, st);
  if (rc) return rc;
  const bool train = (training & 1) != 0, defer = train && (training & PCMI_NET_DEFER_RUNNING_STATS) != 0;
  bool two_seg = false;  // a two-segment pass updates the running estimates through the table as well (segment 0, then 1)
  for (int l = 0; l < n.n_levels; ++l) two_seg |= ps.split[l] < ps.rows[l];
  two_seg &= train;
  ps.upd_n = 0;
  if (defer || two_seg) {
    int n_bn = 0;
    for (int i = 0; i < n_ops; ++i) n_bn += n.ops[i].type == PCMI_OP_BN;
    if (n_bn > ps.upd_cap) {
      PCMI_HIP_CHECK(hipDeviceSynchronize());
      if (ps.upd_host) PCMI_HIP_CHECK(hipHostFree(ps.upd_host));
      if (ps.upd_dev) PCMI_HIP_CHECK(hipFree(ps.upd_dev));
      ps.upd_host = ps.upd_dev = nullptr;
      PCMI_HIP_CHECK(hipHostMalloc((void**)&ps.upd_host, sizeof(BnRunningUpdate) * n_bn, hipHostMallocDefault));
      PCMI_HIP_CHECK(hipMalloc((void**)&ps.upd_dev, sizeof(BnRunningUpdate) * n_bn));
      ps.upd_cap = n_bn;
    }
    // (this wait is also what keeps the enqueueing thread ONE pass ahead of the GPU: host_wait_event sleeps in it)
    if (!ps.upd_copied) PCMI_HIP_CHECK(hipEventCreateWithFlags(&ps.upd_copied, host_wait_event_flags()));
    rc = host_wait_event(ps.upd_copied);  // the previous table has left the pinned buffer
    if (rc) return rc;
  }
  ps.in_feats = in_feats;
  ps.in_ld = in_ld;
  ps.out_feats = out_feats;
  ps.out_ld = out_ld;
  ps.coords = coords;
  // (the pack on a side stream, joined in front of the first convolution that needs it, was measured: 237.7 against
  //  238.3 pairs/s in line -- profiles/r03c_bench_ab.txt -- and is gone)
  if (ps.x3_bwd_pending && ps.x3_bwd_packed) {
    // a training forward of this pass that was never differentiated: its side-stream pack may still be reading the job
    // table / writing the packs this pass is about to rebuild
    PCMI_HIP_CHECK(hipStreamWaitEvent(st, ps.x3_bwd_packed, 0));
    ps.x3_bwd_pending = false;
  }
  const bool train0 = (training & 1) != 0;
  const bool pack_split = train0 && [] {  // PCMI_X3_PACK_SPLIT=0: both orientations at the top of the pass (round 4; A/B)
    const char* e = getenv("PCMI_X3_PACK_SPLIT");
    return !(e && e[0] == '0');
  }();
  // PCMI_X3_PACK_SIDE (round 6, default on for training passes): the forward orientations are packed on the side stream, behind
  // everything enqueued so far, and the pass waits for them in front of the first convolution that reads a pack -- the stem and
  // the 32-channel layers of the first two levels (~0.45 ms) do not.  (Round 5 measured this at +0.3 %, inside the spread, with a
  // join in front of the first convolution of ANY kind; the residual branches above use the same stream.)
  const bool pack_side = train0 && [] {
    const char* e = getenv("PCMI_X3_PACK_SIDE");
    return !(e && e[0] == '0');
  }();
  bool pack_joined = true;
  if (pack_side) {
    rc = ensure_streams(n, false);
    if (rc) return rc;
    if (!n.ev_fwd_fork) PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_fwd_fork, hipEventDisableTiming));
    if (!n.ev_fwd_pack) PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_fwd_pack, hipEventDisableTiming));
    PCMI_HIP_CHECK(hipEventRecord(n.ev_fwd_fork, st));
    PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_fwd_fork, 0));
    rc = x3_prepack(n, ps, params, n.side[0], pack_split);
    PCMI_HIP_CHECK(hipEventRecord(n.ev_fwd_pack, n.side[0]));
    pack_joined = false;
  } else {
    rc = x3_prepack(n, ps, params, st, pack_split);
  }
  const X3TableScope x3_scope;  // the table is this thread's only until the pass has been enqueued
  if (rc) return rc;
  // where the backward-data orientations are packed: behind the first op of the third level (the pass is in its
  // latency-bound coarse phase from there on and the side stream is idle during a forward pass) -- or behind the last op
  // an error return from here on leaves no table behind that a later pass could mistake for this one's packs
  struct X3FailGuard {
    PassState& ps;
    bool armed = true;
    ~X3FailGuard() {
      // (x3_bwd_pending stays: it is true only if THIS pass recorded the event, and the next forward of the pass must still
      //  wait for that side-stream pack before it rebuilds the job table)
      if (armed) ps.x3_current = ps.x3_bwd_owed = false;
    }
  } x3_guard{ps};
  int pack_bwd_at = -1;
  if (ps.x3_bwd_owed) {
    pack_bwd_at = n_ops - 1;
    for (int i = 0; i < n_ops; ++i)
      if (n.tensors[n.ops[i].out].level >= 2) {
        pack_bwd_at = i;
        break;
      }
    rc = ensure_streams(n, false);
    if (rc) return rc;
    if (!ps.x3_bwd_packed) PCMI_HIP_CHECK(hipEventCreateWithFlags(&ps.x3_bwd_packed, hipEventDisableTiming));
  }
  g_prof_fwd.lap(2);
  // PCMI_DEBUG_LATE_WGRAD=<rows> (timing prototype, WRONG results; use with PCMI_DEBUG_SKIP_WGRAD_ROWS=<rows>): the weight
  // gradients of the layers with at least <rows> rows on both sides -- which the backward pass then leaves out -- are
  // launched HERE, on the side stream beside the forward pass, into a scratch buffer (operands: whatever the arenas hold),
  // and the pass waits for them in front of the first such layer: the schedule of "the level-1 weight gradients of
  // iteration i run beside the forward pass of iteration i + 1 and their layers are updated before it reaches them",
  // without the trainer-side plumbing.  profiles/r06b_*.
  int late_first_op = -1;
  if (const long late_rows = debug_env_long("PCMI_DEBUG_LATE_WGRAD"); late_rows > 0 && train0 && ps.grad.p && !ps.grad_off.empty()) {
    rc = ensure_streams(n, false);
    if (rc) return rc;
    static DevBuf late_scratch;
    rc = late_scratch.reserve((size_t)27 * 256 * 256 * sizeof(float), st);
    if (rc) return rc;
    rc = n.ws_side[0].reserve(ps.ws.cap, n.side[0]);
    if (rc) return rc;
    PCMI_HIP_CHECK(hipEventRecord(n.ev_main[0], st));
    PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_main[0], 0));
    const bool late_fp32 = debug_env_long("PCMI_DEBUG_LATE_FP32") != 0;  // the pair-list fp32 kernel for them (small workgroups)
    const char* x3t_env = getenv("PCMI_WGRAD_X3T");
    const std::string x3t_saved = x3t_env ? x3t_env : "";
    if (late_fp32) setenv("PCMI_WGRAD_X3T", "0", 1);
    struct RestoreEnv {
      bool on, had;
      const std::string& v;
      ~RestoreEnv() {
        if (!on) return;
        if (had) setenv("PCMI_WGRAD_X3T", v.c_str(), 1); else unsetenv("PCMI_WGRAD_X3T");
      }
    } restore{late_fp32, x3t_env != nullptr, x3t_saved};
    for (int i = n_ops - 1; i >= 0; --i) {
      const auto& op = n.ops[i];
      if (op.type != PCMI_OP_CONV || op.in == n.input_tensor) continue;
      const int64_t n_in = ps.rows[n.tensors[op.in].level], n_out = ps.rows[n.tensors[op.out].level];
      if (std::min(n_in, n_out) < late_rows) continue;
      late_first_op = i;
      const View x = act_view(n, ps, op.in);
      const View dy = grad_view(n, ps, op.out, out_feats, out_ld);
      rc = spconv_backward_weight(x.p, x.ld, n_in, op.cin, dy.p, dy.ld, n_out, op.cout, ps.has_map[i] ? &ps.maps[i] : nullptr,
                                  op.transpose, (float*)late_scratch.p, nullptr, 0, n.ws_side[0].p, n.ws_side[0].cap, n.side[0]);
      if (rc) return rc;
    }
    if (late_first_op >= 0) PCMI_HIP_CHECK(hipEventRecord(n.ev_side[0], n.side[0]));
  }
  // ---- run --------------------------------------------------------------------------------------
  if (!n.timed_ops.empty()) {  // the next event set of the ring (pcmi_net_time_ops); its old records are dropped
    ++n.timed_cur;
    const size_t per = n.timed_ops.size(), base = (size_t)(n.timed_cur % n.timed_sets) * per;
    for (size_t q = 0; q < per; ++q) n.timed_hit[base + q] = 0;
    if (n.timed_keep_groups) n.timed_group_n[n.timed_cur % n.timed_sets] = 0;
  }
  // ---- the residual branch of a block beside its main path (round 6) ------------------------------------------------------
  // A BasicBlock whose channel count changes computes its residual with a 1x1 convolution + BatchNorm of the block's INPUT
  // (pc/model/resnet.py: downsample); the traced program lists the two behind conv1 / bn1 / conv2 although they depend on
  // nothing the main path computes.  They go to the side stream -- idle during a forward pass but for the weight packs -- as
  // soon as the block's input is complete, with the side workspace, and the BatchNorm that adds the residual waits for them.
  // Same kernels on the same operands: bit-identical results (tests/test_gpu_timing.py).  PCMI_FWD_BRANCH=0: in program order.
  std::vector<int> fork_at(n_ops, -1), join_before(n_ops, -1);
  std::vector<char> on_side(n_ops, 0);
  {
    const char* e = getenv("PCMI_FWD_BRANCH");
    if (!(e && e[0] == '0') && late_first_op < 0) {
      for (int i = 3; i + 2 < n_ops; ++i) {
        const auto &c = n.ops[i], &b1 = n.ops[i + 1], &b2 = n.ops[i + 2];
        if (c.type != PCMI_OP_CONV || c.kernel_size != 1 || b1.type != PCMI_OP_BN || b1.in != c.out || b1.relu || b1.in2 >= 0 ||
            b2.type != PCMI_OP_BN || b2.in2 != b1.out)
          continue;
        int s0 = -1;
        for (int q = i - 3; q < i; ++q)
          if (n.ops[q].in == c.in && !on_side[q]) {
            s0 = q;
            break;
          }
        if (s0 < 0 || fork_at[s0] >= 0) continue;
        fork_at[s0] = i;
        on_side[i] = on_side[i + 1] = 1;
        join_before[i + 2] = i;
      }
    }
  }
  bool any_fork = false;
  for (int i = 0; i < n_ops; ++i) any_fork |= fork_at[i] >= 0;
  if (any_fork) {
    rc = ensure_streams(n, false);
    if (rc) return rc;
    rc = n.ws_side[0].reserve(ps.ws.cap, n.side[0]);
    if (rc) return rc;
    if (!n.ev_fwd_fork) PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_fwd_fork, hipEventDisableTiming));
    if (!n.ev_fwd_join) PCMI_HIP_CHECK(hipEventCreateWithFlags(&n.ev_fwd_join, hipEventDisableTiming));
  }
  auto run_op = [&](int i, hipStream_t sq, DevBuf& w) -> int {
    const auto& op = n.ops[i];
    const View x = act_view(n, ps, op.in), y = act_view(n, ps, op.out);
    const int64_t n_in = ps.rows[n.tensors[op.in].level], n_out = ps.rows[n.tensors[op.out].level];
    int rc = PCMI_OK;
      const int tq = n.timed_slot(i);
      if (tq >= 0) PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tq + 0], sq));
      const long lf0 = g_launches;
      if (op.type == PCMI_OP_CONV) {
        rc = spconv_forward(x.p, x.ld, n_in, op.cin, params + op.w_off, op.cout, ps.has_map[i] ? &ps.maps[i] : nullptr,
                            op.transpose, op.has_bias ? params + op.b_off : nullptr, y.p, y.ld, n_out, 0, w.p, w.cap,
                            sq);
      } else if (op.type == PCMI_OP_BN) {
        View r = {nullptr, 0};
        if (op.in2 >= 0) r = act_view(n, ps, op.in2);
        float* stats0 = (float*)(ps.act.p + ps.stat_off[i]);
        uint32_t* rbits = ps.bits_off[i] != SIZE_MAX ? (uint32_t*)(ps.act.p + ps.bits_off[i]) : nullptr;
        if (train) {
          const int64_t sp = ps.split[n.tensors[op.in].level];
          float* stats1 = stats0 + 3 * op.cout;
          if (sp < n_in) {  // both segments in one statistics launch + one apply launch; running estimates via the table
            rc = bn_forward_train2(x.p, x.ld, n_in, sp, op.cout, params + op.w_off, params + op.b_off, op.eps, r.p, r.ld, op.relu,
                                   y.p, y.ld, stats0, stats0 + op.cout, stats0 + 2 * op.cout, 3 * op.cout, w.p, w.cap, sq, rbits);
            if (op.running_mean)
              ps.upd_host[ps.upd_n++] = {op.running_mean, op.running_var, stats0, stats0 + 2 * op.cout, op.cout, op.momentum,
                                         stats1, stats1 + 2 * op.cout};
          } else {
            const bool tab = defer || two_seg;
            rc = bn_forward_train(x.p, x.ld, n_in, op.cout, params + op.w_off, params + op.b_off,
                                  tab ? nullptr : op.running_mean, tab ? nullptr : op.running_var, op.momentum, op.eps, r.p,
                                  r.ld, op.relu, y.p, y.ld, stats0, stats0 + op.cout, stats0 + 2 * op.cout, w.p, w.cap, sq, rbits);
            if (tab && op.running_mean)
              ps.upd_host[ps.upd_n++] = {op.running_mean, op.running_var, stats0, stats0 + 2 * op.cout, op.cout, op.momentum,
                                         nullptr, nullptr};
          }
        } else {
          rc = pcmi_bn_fwd_eval(x.p, x.ld, n_in, op.cout, params + op.w_off, params + op.b_off, op.running_mean,
                                op.running_var, op.eps, r.p, r.ld, op.relu, y.p, y.ld, (void*)sq);
        }
      } else {
        rc = pcmi_l2norm_fwd(x.p, x.ld, n_in, op.cout, y.p, y.ld, (float*)(ps.act.p + ps.stat_off[i]), (void*)sq);
      }
      if (rc) return rc;
      if (tq >= 0) {
        PCMI_HIP_CHECK(hipEventRecord(n.timed_ev[pcmi_net::kTimedEv * tq + 1], sq));
        n.timed_hit[tq] |= 1;
        n.timed_cnt[3 * tq + 0] = (int)(g_launches - lf0);
      }
    return PCMI_OK;
  };
  for (int i = 0; i < n_ops; ++i) {
    const auto& op = n.ops[i];
    if (i == late_first_op) PCMI_HIP_CHECK(hipStreamWaitEvent(st, n.ev_side[0], 0));  // (PCMI_DEBUG_LATE_WGRAD)
    if (!pack_joined && (i >= ps.x3_first_op || fork_at[i] >= 0)) {  // (a forked 1x1 layer may pack for itself on the side stream: keep it simple)
      PCMI_HIP_CHECK(hipStreamWaitEvent(st, n.ev_fwd_pack, 0));
      pack_joined = true;
    }
    if (fork_at[i] >= 0) {  // everything enqueued so far is the branch's input (and more): the branch starts behind it
      PCMI_HIP_CHECK(hipEventRecord(n.ev_fwd_fork, st));
      PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_fwd_fork, 0));
      rc = run_op(fork_at[i], n.side[0], n.ws_side[0]);
      if (rc) return rc;
      rc = run_op(fork_at[i] + 1, n.side[0], n.ws_side[0]);
      if (rc) return rc;
      PCMI_HIP_CHECK(hipEventRecord(n.ev_fwd_join, n.side[0]));
    }
    if (join_before[i] >= 0) PCMI_HIP_CHECK(hipStreamWaitEvent(st, n.ev_fwd_join, 0));
    if (!on_side[i]) {
      rc = run_op(i, st, ps.ws);
      if (rc) return rc;
    }
    if (i == pack_bwd_at) {  // the side stream starts behind this point of the pass (and behind the job table's upload)
      PCMI_HIP_CHECK(hipEventRecord(n.ev_main[0], st));
      PCMI_HIP_CHECK(hipStreamWaitEvent(n.side[0], n.ev_main[0], 0));
      rc = x3_pack_many(reinterpret_cast<const X3PackJob*>(ps.x3_jobs_dev.p), ps.x3_n_jobs, ps.x3_items, n.side[0], ps.x3_items_fwd);
      if (rc) return rc;
      PCMI_HIP_CHECK(hipEventRecord(ps.x3_bwd_packed, n.side[0]));
      ps.x3_bwd_owed = false;
      ps.x3_bwd_pending = true;
    }
    g_prof_fwd.lap(op.type == PCMI_OP_CONV ? 3 : (op.type == PCMI_OP_BN ? 4 : 5));
  }
  if ((defer || two_seg) && ps.upd_n > 0) {
    PCMI_HIP_CHECK(hipMemcpyAsync(ps.upd_dev, ps.upd_host, sizeof(BnRunningUpdate) * ps.upd_n, hipMemcpyHostToDevice, st));
    PCMI_HIP_CHECK(hipEventRecord(ps.upd_copied, st));
    if (!defer) {  // two-segment pass without a deferral request: apply now, in layer order
      rc = bn_running_update(ps.upd_dev, ps.upd_n, st);
      if (rc) return rc;
      ps.upd_n = 0;
    }
  }
  g_prof_fwd.lap(6);
  static const char* const kFwdNames[] = {"levels", "maps", "layout+reserve", "conv", "bn", "l2norm", "tail"};
  g_prof_fwd.report("net_forward", kFwdNames, 7);
  ps.valid = train;
  x3_guard.armed = false;
  return PCMI_OK;
}

int pcmi_net_stream_wait_bucket(pcmi_net_t* net, pcmi_stream_t stream) {
  PCMI_REQUIRE(net, PCMI_ERR_INVALID, "net_stream_wait_bucket: null argument");
  PCMI_REQUIRE(net->bkt_valid, PCMI_ERR_INVALID, "net_stream_wait_bucket: only valid inside a pcmi_ready_fn callback");
  PCMI_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), net->ev_bkt_main, 0));
  if (!debug_env_long("PCMI_DEBUG_SKIP_BUCKET_SIDE_WAIT"))
    for (int q = 0; q < pcmi_net::kSides; ++q)
      if (net->bkt_side[q]) PCMI_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), net->ev_bkt_side[q], 0));
  return PCMI_OK;
}

int pcmi_net_time_ops(pcmi_net_t* net, const int* ops, int n_ops, int n_sets) {
  PCMI_REQUIRE(net && (n_ops == 0 || (ops && n_sets > 0)) && n_ops >= 0 && n_ops <= 1024 && n_sets <= 64, PCMI_ERR_INVALID,
               "net_time_ops: bad argument");
  PCMI_HIP_CHECK(hipDeviceSynchronize());  // nothing in flight records into the events that go away
  net->timed_clear();
  if (n_ops == 0) return PCMI_OK;
  for (int q = 0; q < n_ops; ++q)
    PCMI_REQUIRE(ops[q] >= 0 && ops[q] < (int)net->ops.size(), PCMI_ERR_INVALID, "net_time_ops: op %d is not an op of this network", ops[q]);
  net->timed_ops.assign(ops, ops + n_ops);
  net->timed_sets = n_sets;
  net->timed_hit.assign((size_t)n_sets * n_ops, 0);
  net->timed_cnt.assign((size_t)3 * n_sets * n_ops, 0);
  for (int e = 0; e < pcmi_net::kTimedEv * n_sets * n_ops; ++e) {
    hipEvent_t ev = nullptr;
    PCMI_HIP_CHECK(hipEventCreate(&ev));
    net->timed_ev.push_back(ev);
  }
  return PCMI_OK;
}

int pcmi_net_time_all(pcmi_net_t* net, int n_sets) {
  PCMI_REQUIRE(net && n_sets >= 0 && n_sets <= 64, PCMI_ERR_INVALID, "net_time_all: bad argument");
  if (n_sets == 0) return pcmi_net_time_ops(net, nullptr, 0, 0);
  std::vector<int> all(net->ops.size());
  for (size_t i = 0; i < all.size(); ++i) all[i] = (int)i;
  const int rc = pcmi_net_time_ops(net, all.data(), (int)all.size(), n_sets);
  if (rc) return rc;
  net->timed_keep_groups = true;
  net->timed_group_n.assign(n_sets, 0);
  net->timed_group_cnt.assign((size_t)pcmi_net::kTimedGroups * n_sets, 0);
  for (int e = 0; e < 2 * pcmi_net::kTimedGroups * n_sets; ++e) {
    hipEvent_t ev = nullptr;
    PCMI_HIP_CHECK(hipEventCreate(&ev));
    net->timed_group_ev.push_back(ev);
  }
  return PCMI_OK;
}

int pcmi_net_timed_groups_ms(pcmi_net_t* net, int set, float* ms, int cap, int* n_out) {
  PCMI_REQUIRE(net && ms && n_out && cap >= 0 && net->timed_keep_groups && set >= 0 && set < net->timed_sets, PCMI_ERR_INVALID,
               "net_timed_groups_ms: bad argument (or pcmi_net_time_all is not active)");
  const int n = std::min(cap, net->timed_group_n[set]);
  for (int g = 0; g < n; ++g) {
    const int slot = set * pcmi_net::kTimedGroups + g;
    PCMI_HIP_CHECK(hipEventSynchronize(net->timed_group_ev[2 * slot + 1]));
    PCMI_HIP_CHECK(hipEventElapsedTime(&ms[g], net->timed_group_ev[2 * slot], net->timed_group_ev[2 * slot + 1]));
  }
  *n_out = n;
  return PCMI_OK;
}

int pcmi_net_timed_launches(pcmi_net_t* net, int set, int* fwd, int* bwd, int* wgrad, int n_ops, int* groups, int groups_cap) {
  PCMI_REQUIRE(net && fwd && bwd && wgrad && n_ops == (int)net->timed_ops.size() && set >= 0 && set < net->timed_sets,
               PCMI_ERR_INVALID, "net_timed_launches: bad argument");
  for (int q = 0; q < n_ops; ++q) {
    const size_t h = (size_t)set * n_ops + q;
    fwd[q] = (net->timed_hit[h] & 1) ? net->timed_cnt[3 * h + 0] : 0;
    bwd[q] = (net->timed_hit[h] & 2) ? net->timed_cnt[3 * h + 1] : 0;
    wgrad[q] = (net->timed_hit[h] & 4) ? net->timed_cnt[3 * h + 2] : 0;
  }
  if (groups && net->timed_keep_groups)
    for (int g = 0; g < groups_cap; ++g)
      groups[g] = g < net->timed_group_n[set] ? net->timed_group_cnt[(size_t)set * pcmi_net::kTimedGroups + g] : 0;
  return PCMI_OK;
}

int pcmi_net_timed_ms(pcmi_net_t* net, int set, float* fwd_ms, float* bwd_ms, float* wgrad_ms, int n_ops) {
  PCMI_REQUIRE(net && fwd_ms && bwd_ms && wgrad_ms && n_ops == (int)net->timed_ops.size() && set >= 0 && set < net->timed_sets,
               PCMI_ERR_INVALID, "net_timed_ms: set %d / %d slots asked, %d sets of %d ops are timed", set, n_ops,
               net ? net->timed_sets : 0, net ? (int)net->timed_ops.size() : 0);
  constexpr int E = pcmi_net::kTimedEv;
  for (int q = 0; q < n_ops; ++q) {
    const size_t h = (size_t)set * n_ops + q;
    float* dst[3] = {&fwd_ms[q], &bwd_ms[q], &wgrad_ms[q]};
    for (int kind = 0; kind < 3; ++kind) {
      *dst[kind] = -1.f;  // not recorded in this set (or no backward-data: the op reads the network input)
      if (net->timed_hit[h] & (1 << kind)) {
        PCMI_HIP_CHECK(hipEventSynchronize(net->timed_ev[E * h + 2 * kind + 1]));
        PCMI_HIP_CHECK(hipEventElapsedTime(dst[kind], net->timed_ev[E * h + 2 * kind], net->timed_ev[E * h + 2 * kind + 1]));
      }
    }
  }
  return PCMI_OK;
}

int pcmi_net_apply_running_stats(pcmi_net_t* net, int pass, pcmi_stream_t stream) {
  PCMI_REQUIRE(net && pass >= 0 && pass < (int)net->passes.size(), PCMI_ERR_INVALID, "net_apply_running_stats: bad argument");
  PassState& ps = net->passes[pass];
  const int n_upd = ps.upd_n;
  ps.upd_n = 0;  // applied once
  return bn_running_update(ps.upd_dev, n_upd, as_stream(stream));
}

int pcmi_net_backward(pcmi_net_t* net, int pass, const float* d_out, int64_t d_ld, const float* params, float* grads,
                      const int64_t* bucket_lo_host, int n_buckets, pcmi_ready_fn ready, void* ready_ctx,
                      pcmi_stream_t stream) {
  PCMI_REQUIRE(net && d_out && params && grads, PCMI_ERR_INVALID, "net_backward: null argument");
  PCMI_REQUIRE(pass >= 0 && pass < (int)net->passes.size() && net->passes[pass].valid, PCMI_ERR_INVALID,
               "net_backward: pass %d has no training-mode forward to differentiate", pass);
  BackwardJob job;
  job.pass = pass;
  job.d_out = d_out;
  job.d_ld = d_ld;
  job.st = as_stream(stream);
  return run_backward(*net, job, params, grads, bucket_lo_host, n_buckets, ready, ready_ctx);
}

}  // extern "C"
