// shoulder_hip.hip -- libshoulder_hip.so: context, buffers, C-ABI (include/shoulder_hip.h) and the
// stage runner.  gfx950 only.  Kernels live in k_*.h next to this file.
#include "../../include/shoulder_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <vector>

#include "k_slices.h"
#include "k_ovf.h"
#include "k_stages.h"
#include "k_groove.h"
#include "k_anp.h"
#include "k_unet.h"
#include "k_unet_bf16.h"
#include "k_unet16_ldr.h"
#include "k_unet_x3.h"
#include "k_unet16_up.h"
#include "unet16_pp.h"
#include "k_stl.h"
#include "k_clip.h"
#include "k_hullpre.h"
#include "k_te.h"
#include "k_obb.h"
#include "k_hull.h"
#include "sh_hull.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <sched.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types for sh_comm.h; librccl itself is bound at run time

using namespace sh;

// ---------------------------------------------------------------------------------------------
struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
  int elem = 1;
  size_t per_mesh = 0;     // bytes per humerus for [B][...] buffers (0: shared / ragged / scratch)
};

struct KTimer {
  double ms = 0;
  int n = 0;
};

struct sh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  sh_params params;
  int B = 0;
  long long sumV = 0, sumF = 0, maxV = 0, maxF = 0;
  std::vector<long long> h_voff, h_foff;
  std::map<std::string, Buf> bufs;
  bool have_rfc = false, have_unet = false;
  int rfc_nodes = 0, rfc_trees = 0;
  int unet_base = 0, unet_depth = 0;
  std::vector<float> h_unet;                 // packed UNet parameters (host copy)
  std::vector<int32_t> h_feat, h_ti, h_fi, h_roots;
  std::vector<float> h_thr, h_lw;
  struct ULayer { size_t w_off, b_off; int cin, cout, taps; };
  std::map<std::string, ULayer> ulayers;
  size_t unet_floats = 0;
  bool obb_injected = false;
  void* comm = nullptr;                      // sh_comm_init_all: this context's RCCL communicator (sh_comm.h), its rank and the group's size
  int comm_rank = -1, comm_n = 0;
  // sh_set_keep_products: every plane's resampled contour and polar rows leave k_resample_polar (k_slices.h, RsWant); off: the rows
  // the later stages read.  rs_*: what the last SH_STAGE_PROXIMAL run of the resident batch wrote (SH_STAGE_GROOVE checks it covers its rows)
  bool keep_products = false;
  const double* unet_raw = nullptr;          // run_window -> unet_forward16: the unscaled image and its encoded range, when the first kernel scales it itself
  const unsigned long long* unet_mm = nullptr;
  int rs_cs_lo = 0, rs_cs_hi = 0;
  bool rs_all = false;
  unsigned long long rs_gen = ~0ull;
  int rec_rows = 0;                          // sh_set_record_rows: 0 = full sh_landmarks records, R > 0 = packed records with R anatomic-neck rows
  bool bounds_cleared = false;               // run_obb's first fill of this window covered zb_enc / anp.mm_enc (run_window skips its own)
  // hull of SH_STAGE_OBB: 1 = on the device (k_hull.h), 0 = host quickhull (sh_hull.h).  sh_set_hull_mode / SHOULDER_HULL=host|device|auto.
  // A humerus the device hull gives up (pinched horizon on nearly coplanar clouds, capacities) is re-done ALONE by sh_collect:
  // host quickhull for that humerus, its record patched into the device buffers, its stages re-run as a window of one behind
  // whatever else is in flight on the stream (redo_given_up).  `hulld.skip[b]` then keeps the device hull off that humerus for
  // as long as the batch stays resident (skip_gen == batch_gen).
  int hull_mode = 1;
  unsigned long long skip_gen = ~0ull;
  int skip_nfmax = 0;                      // most hull faces among the humeri of this batch that are on the host hull (hulld.skip)
  // overflow pools of the slice layer (k_ovf.h): capacities in segments / ring points / bytes; grown by sh_collect on demand
  unsigned long long ovf_seg_cap = 1ull << 18, ovf_ring_cap = 1ull << 18, ovf_work_cap = 32ull << 20;
  // a run of the resident batch that planned no overflow plane in any set (ctr[4] == 0 at collect) lets later runs of the SAME batch
  // and parameters skip the overflow tier's launches (they would all return at once: ~17 launches, ~60 us per step)
  unsigned long long ovf_none_gen = ~0ull;
  int end_cap = SH_ENDCAP;                   // points per end section "obb.endpts" holds (grown by sh_collect like the pools)
  unsigned long long obb_gen = ~0ull;        // the batch generation the three fields below belong to
  HullCap hcap = {SH_HV, SH_HF, SH_HE};      // per-humerus capacity (= stride) of the hull record and the per-face obb.* arrays; a batch with a larger
                                             // hull grows it (grow_hull_records) -- every kernel takes the strides as an argument
  int obb_sil_need = 0;                      // the longest silhouette (edges) a direction of the resident batch had when it overflowed a tier of
                                             // k_obb_candidates: later runs take the tier that holds it (reset with the batch)
  bool obb_nf_over = false;                  // a device-hull run met a hull with more faces than its candidates tier masks: the next run takes the workspace tier
  bool hull_force_host = false;              // the resident batch has a hull above the device hull's record: its hulls come from the host (reset with the batch)
  bool redo_records = false;               // run_obb: the hull records of the window are in place already (redo_given_up)
  int redo_nf = 0;
  std::vector<float> h_verts;                // host copy of the vertices (hull stage)
  bool h_verts_valid = false;
  // device-generated batches: the hull's points come back through the prefilter (k_hullpre.h) into pinned memory
  float* h_kept = nullptr; long long h_kept_cap = 0;
  int* h_nkept = nullptr; int h_nkept_cap = 0;
  struct HullPts { const float* src = nullptr;            // what hull_host_phase reads: h_verts.data() or the pinned survivors
                   std::vector<long long> off;            // first point of humerus b in src
                   std::vector<int> cnt; };               // points of humerus b in src
  HullPts hull_in;                           // ... of the resident batch
  long long* h_koff = nullptr;               // pinned: offsets of the survivors (B + 1)
  // The STAGING SIDE of the mesh slot (sh_stage_meshes / sh_stage_stl / sh_commit_staged): the next batch is copied into buffers of
  // its own ("verts.s", "faces.s", "voff.s", "foff.s") on the copy stream while a run of the resident batch executes, its hull
  // points come back through a prefilter scratch of its own ("hullpre.*.s") and its hulls are computed by the background thread
  // (`prep`, gen = batch_gen + 1) -- sh_commit_staged then only swaps the buffer entries and the next sh_submit finds its hulls.
  struct StageSide {
    bool active = false, from_stl = false;
    int B = 0; long long sumV = 0, sumF = 0, maxV = 0, maxF = 0;
    std::vector<long long> voff, foff;
    void* h_src = nullptr; size_t h_src_cap = 0;            // pinned staging of the caller's arrays / files
    int* h_flag = nullptr;                                  // pinned: validation word (+ STL: counts and non-finite words behind it)
    size_t h_flag_cap = 0;
    float* h_kept = nullptr; long long h_kept_cap = 0;      // pinned: prefilter survivors of the staged batch
    long long* h_koff = nullptr; int h_koff_cap = 0;
    HullPts pts;
    hipEvent_t ready_ev = nullptr;                          // everything the commit needs is on the device
    std::mutex m; std::condition_variable cv; bool meta_ready = true; int meta_rc = 0; std::string meta_err;      // STL: sizes known
  } stg;
  // Window of the batch the stage runner is working on: sh_run walks the batch in windows so that the
  // host hull of window k+1 overlaps the device work of window k.  buf<T>() applies the offset.
  int b0 = 0, Bwin = 0;
  struct HullStage { double* hv = nullptr; double* nr = nullptr; int* ed = nullptr; int* cnt = nullptr; int cap = 0; hipEvent_t ev = nullptr; bool used = false;
                     int pv = 4096, pf = 8192, pe = 12288; };      // per-humerus pitch of the pinned staging (elements): the usual hull fits the small
                                                                 // one; a batch with a larger hull re-allocates the slot at SH_HV / SH_HF / SH_HE
  HullStage hstage[2];                       // pinned host staging, double buffered
  int hslot = 0;                             // slot the next hull goes to
  // Overlap (sh_set_overlap): while the device works on run k, a background thread computes the hulls run k+1 will
  // need (same resident batch -- invalidated by any upload) into the other pinned slot.
  struct Prepared {
    std::thread th; bool active = false; int slot = 0, B = 0, rc = SH_OK, bad_mesh = -1; unsigned long long gen = 0;
    double d2h_ms = 0, hull_ms = 0; std::string err;
    bool uploaded = false;      // the hull records are already in the device buffers (copied by the background thread)
    bool staged = false;        // the thread works for the STAGED batch (gen = the generation the batch gets at sh_commit_staged)
  } prep;
  hipEvent_t obb_done_ev = nullptr;      // recorded after the last kernel of a run that reads the hull.* device buffers
  // sh_submit / sh_collect: up to two runs in flight (the second one is enqueued while the first still executes)
  struct Ticket { hipEvent_t ev = nullptr; int* h_err = nullptr; int* h_fail = nullptr; unsigned long long* h_ovf = nullptr; int cap = 0, B = 0; bool pending = false; sh_landmarks* host_out = nullptr;
                  uint32_t mask = 0; sh_landmarks* out_arg = nullptr; bool dev_hull = false; unsigned long long gen = 0; size_t rec = sizeof(sh_landmarks); int rows = 0; };
  Ticket tickets[2];
  int t_head = 0, t_tail = 0, n_pending = 0;
  hipStream_t out_stream = nullptr;      // sh_collect copies the records / status words of a finished run to the host on this stream
  bool overlap = false;
  bool unet_turn = false;                // sh_set_unet_turns: UNet passes of the contexts of one device run one after another
  hipEvent_t unet_done_ev = nullptr;
  unsigned long long batch_gen = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t stl_counted_ev = nullptr;      // sh_stage_stl: the device has counted the merged vertices / faces
  // side stream of the stage runner: the distal slice set and the rectangles of the trans-epicondylar stage hang on nothing but the
  // box frame, so they run beside the full -> neck -> proximal chain (SHOULDER_SIDE_STREAM=0: everything on the one stream)
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork_ev = nullptr, side_join_ev = nullptr;
  bool side_pending = false;
  // timing
  bool zero_page_ready = false;
  // Switches of equivalent paths (the A/B arms of tests/), read from the environment ONCE, when the context is created -- no launch
  // path consults the environment.
  struct Switches {
    int window = 0;            // SHOULDER_WINDOW=n: humeri per window of the host-hull walk (0: SH_WINDOW)
    bool obb_prune = true;     // SHOULDER_OBB_PRUNE=0: every hull-face direction is evaluated
    bool slice_merge = true;   // SHOULDER_SLICE_MERGE=0: one slice set per launch group
    bool side_stream = true;   // SHOULDER_SIDE_STREAM=0: small batches keep the distal branch in the chain
    bool up_inside = false;    // SHOULDER_UP_INSIDE=1: up1 inside dec1a's loader waves instead of a launch of its own (k_upconv16g).  Off: the network
                               // alone is 0.07 ms faster with it, the two-lane step 0.1-0.2 ms slower
    int te_early = -1;         // SHOULDER_TE_EARLY=1: the trans-epicondylar part forked beside the lane's UNet; 0: behind the UNet; unset: in front of it
    bool debug = false;        // SH_DEBUG: host-phase timings on stderr
  } sw;
  bool unet_reference = false;      // SHOULDER_UNET_REFERENCE=1 at context creation: the 16-bit network layer by layer on the generic kernels
  int ticket_next = 0;              // next free work counter of "unet16.tickets" (one per persistent conv launch of a forward pass)
  std::map<std::tuple<int, int, int>, std::pair<int, int>> tk_tabs;      // (items, workgroups, cout groups) -> (offset, tickets) in "unet16.tk_tab"
  int tk_tab_used = 0;
  bool packtab_ready = false;      // layer table of k_pack_w_bf16_all uploaded (reset by sh_load_unet)
  bool packed_rfc = false;         // "rfc.nodes" holds the packed forest of the CURRENT parameter block
  bool packed_x3 = false;          // "params_x3h/l" hold the split weights of the CURRENT parameter block (reset with packed_kind)
  int packed_kind = -1;            // element kind (0 bf16, 1 f16) "params_bf16" was packed for from the CURRENT parameter block; -1: repack.
                                   // Reset wherever the block can change: sh_load_*, sh_param_block (the pointer goes to the caller), sh_param_block_commit
  int num_cus = 0;
  int timing = 0;      // 0 off, 1 every launch, 2 UNet layers only
  std::vector<std::tuple<std::string, hipEvent_t, hipEvent_t>> pending;
  std::map<std::string, KTimer> timers;
};

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return SH_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

static int fail(sh_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

static int ensure(sh_ctx* c, const char* name, size_t bytes, int elem, void** out = nullptr) {
  Buf& b = c->bufs[name];
  if (b.bytes < bytes || b.p == nullptr) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    hipError_t e = hipMalloc(&b.p, bytes ? bytes : 16);
    if (e != hipSuccess) {
      c->err = std::string("hipMalloc(") + name + "): " + hipGetErrorString(e);
      b.bytes = 0;
      return SH_ERR_NOMEM;
    }
    b.bytes = bytes;
  }
  b.elem = elem;
  if (out) *out = b.p;
  return SH_OK;
}

template <typename T>
static T* buf(sh_ctx* c, const char* name) {
  auto it = c->bufs.find(name);
  if (it == c->bufs.end() || !it->second.p) return nullptr;
  return (T*)((char*)it->second.p + (size_t)c->b0 * it->second.per_mesh);
}

// kernel launch with optional HIP-event timing on the ctx stream
// timing level 1: events around every launch; 2: around the UNet layers only ("unet.*": ~25 launches per run, so the
// measurement does not stretch the run it measures -- events around all ~150 launches cost ~0.7 ms per run at B = 64)
static inline bool timed_launch(const sh_ctx* c, const char* name) {
  return c->timing == 1 || (c->timing == 2 && name[0] == 'u' && name[1] == 'n' && name[2] == 'e' && name[3] == 't' && name[4] == '.');
}

#define LAUNCH(ctx, name, kernel, grid, block, ...)                                         \
  do {                                                                                      \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                \
    const bool timed_ = timed_launch(ctx, name);                                            \
    if (timed_) {                                                                           \
      (void)hipEventCreate(&e0_); (void)hipEventCreate(&e1_);                               \
      (void)hipEventRecord(e0_, (ctx)->stream);                                             \
    }                                                                                       \
    hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);                 \
    if (timed_) {                                                                           \
      (void)hipEventRecord(e1_, (ctx)->stream);                                             \
      (ctx)->pending.emplace_back(name, e0_, e1_);                                          \
    }                                                                                       \
    HIPCHK(ctx, hipGetLastError());                                                         \
  } while (0)

// the same bookkeeping around a launcher function of another translation unit (unet16_pp.h)
#define LAUNCH_FN(ctx, name, call)                                                           \
  do {                                                                                      \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                \
    const bool timed_ = timed_launch(ctx, name);                                            \
    if (timed_) {                                                                           \
      (void)hipEventCreate(&e0_); (void)hipEventCreate(&e1_);                               \
      (void)hipEventRecord(e0_, (ctx)->stream);                                             \
    }                                                                                       \
    call;                                                                                   \
    if (timed_) {                                                                           \
      (void)hipEventRecord(e1_, (ctx)->stream);                                             \
      (ctx)->pending.emplace_back(name, e0_, e1_);                                          \
    }                                                                                       \
    HIPCHK(ctx, hipGetLastError());                                                         \
  } while (0)

// Buffer clears of a run as ONE kernel launch per group of adjacent clears.  hipMemsetAsync costs the enqueuing thread ~60 us per
// call on this stack (rocprofv3 trace of bench.py: the five clears that open a step spread over 0.3 ms before its first kernel;
// ~17 per step = 1 ms of host time) -- a kernel launch costs ~5 us.
struct FillList { void* p[6]; unsigned long long n[6]; unsigned v[6]; };
__global__ void k_fill_list(FillList L) {
  unsigned* p = (unsigned*)L.p[blockIdx.y];
  const unsigned long long nw = L.n[blockIdx.y] >> 2;
  const unsigned v = L.v[blockIdx.y];
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < nw; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v;
}
struct FillEnt { void* p; size_t bytes; unsigned char byte; };      // bytes: a multiple of 4, p 4-byte aligned
static int fill_list(sh_ctx* c, std::initializer_list<FillEnt> ents) {
  FillList L{};
  int k = 0; size_t nmax = 0;
  for (const FillEnt& e : ents) {
    if (!e.p || e.bytes == 0) continue;
    if (k == 6 || (e.bytes & 3) || ((uintptr_t)e.p & 3)) return fail(c, SH_ERR_STATE, "fill_list: bad entry");
    L.p[k] = e.p; L.n[k] = e.bytes; L.v[k] = 0x01010101u * e.byte; ++k;
    nmax = std::max(nmax, e.bytes);
  }
  if (k == 0) return SH_OK;
  const unsigned gx = (unsigned)std::min<size_t>(512, std::max<size_t>(1, (nmax / 4 + 256 * 16 - 1) / (256 * 16)));
  LAUNCH(c, "fill", k_fill_list, dim3(gx, (unsigned)k), dim3(256), L);
  return SH_OK;
}
#define FILL(ctx, ...) do { int frc_ = fill_list(ctx, {__VA_ARGS__}); if (frc_ != SH_OK) return frc_; } while (0)

static void drain_timers(sh_ctx* c) {
  for (auto& t : c->pending) {
    float ms = 0;
    (void)hipEventSynchronize(std::get<2>(t));
    if (hipEventElapsedTime(&ms, std::get<1>(t), std::get<2>(t)) == hipSuccess) {
      KTimer& k = c->timers[std::get<0>(t)];
      k.ms += ms;
      k.n += 1;
    }
    (void)hipEventDestroy(std::get<1>(t));
    (void)hipEventDestroy(std::get<2>(t));
  }
  c->pending.clear();
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int sh_default_params(sh_params* p) {
  if (!p) return SH_ERR_ARG;
  p->canal_cutoff[0] = 0.35; p->canal_cutoff[1] = 0.75;
  p->groove_cutoff[0] = 0.2; p->groove_cutoff[1] = 0.75;
  p->groove_deg_window = 7.0;
  p->unet_dtype = SH_UNET_F32;
  p->bone_kind = SH_BONE_HUMERUS;
  return SH_OK;
}

// "host" | "device" | "auto" (default).  auto: the host quickhull while this rank has enough usable hardware threads to itself -- 16 when it is
// the only rank of its host, 48 per rank otherwise -- (it
// is free for the GPU and hidden behind the previous step: at ~7 000 humeri/s a rank keeps ~14 cores busy with hulls), the device
// hull otherwise -- 8 ranks on a 256-thread host, a thin host, a rank pinned to a few cores (affinity mask), a cgroup CPU quota.
// Hardware threads this process may actually use: its affinity mask (taskset, a pinned rank, a container's cpuset) capped by a
// cgroup-v2 CPU quota where one is set (cpu.max "<quota> <period>") -- std::thread::hardware_concurrency() sees neither.
static unsigned usable_threads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) n = (unsigned)k; }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0}; long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    }
    fclose(f);
  }
  return n;
}

static unsigned affinity_threads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) n = (unsigned)k; }
  return n;
}

// `sustained`: count a CPU quota (it caps the AVERAGE cpu time -- what decides whether a rank can afford host hulls at all); the size
// of the worker pool goes by the affinity mask alone: a quota does not stop 32 threads from running a 4 ms burst side by side
// (measured under a 16-CPU quota: hull phase of a 64-batch 4.1 ms with 32 workers, 6.3 ms with 16).
static unsigned threads_per_local_rank(bool sustained = true) {
  unsigned hw = sustained ? usable_threads() : affinity_threads();
  if (const char* w = getenv("LOCAL_WORLD_SIZE")) { int v = atoi(w); if (v > 1) hw = std::max(1u, hw / (unsigned)v); }
  return hw;
}

static int hull_mode_from(const char* e) {
  if (e && (e[0] == 'h' || e[0] == '0')) return 0;
  if (e && (e[0] == 'd' || e[0] == '1')) return 1;
  // A rank keeps ~14 cores busy with hulls at ~7 000 humeri/s.  Several ranks on one host: host hull only with >= 48 threads per rank
  // (the ranks' pools and submitting threads must not fight for the cores).  A single rank: >= 16 usable threads are enough --
  // measured under a 16-CPU cgroup quota, B = 64, 20 steps: host hull 7 060 humeri/s (16 workers), device hull 6 720.
  const char* w = getenv("LOCAL_WORLD_SIZE");
  const bool alone = !(w && atoi(w) > 1);
  return threads_per_local_rank() >= (alone ? 16u : 48u) ? 0 : 1;
}

int sh_set_hull_mode(sh_ctx* c, const char* mode) {
  if (!c || !mode) return SH_ERR_ARG;
  if (strcmp(mode, "host") && strcmp(mode, "device") && strcmp(mode, "auto")) return fail(c, SH_ERR_ARG, "sh_set_hull_mode: host | device | auto");
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_set_hull_mode: runs are in flight");
  if (c->prep.active) { if (c->prep.th.joinable()) c->prep.th.join(); c->prep.active = false; c->prep.gen = ~0ull; }
  c->hull_mode = hull_mode_from(strcmp(mode, "auto") ? mode : nullptr);
  return SH_OK;
}

int sh_get_hull_mode(const sh_ctx* c) { return c ? c->hull_mode : SH_ERR_ARG; }
int sh_auto_hull_mode(void) { return hull_mode_from(nullptr); }

int sh_ctx_create(int device, void* hip_stream, sh_ctx** out) {
  if (!out) return SH_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SH_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return SH_ERR_HIP;
  sh_ctx* c = new (std::nothrow) sh_ctx();
  if (!c) return SH_ERR_NOMEM;
  c->device = device;
  c->hull_mode = hull_mode_from(getenv("SHOULDER_HULL"));
  c->unet_reference = getenv("SHOULDER_UNET_REFERENCE") && getenv("SHOULDER_UNET_REFERENCE")[0] == '1';
  {
    auto off = [](const char* n) { const char* e = getenv(n); return e && e[0] == '0'; };
    if (const char* e = getenv("SHOULDER_WINDOW")) { const int v = atoi(e); if (v > 0) c->sw.window = v; }
    c->sw.obb_prune = !off("SHOULDER_OBB_PRUNE");
    c->sw.slice_merge = !off("SHOULDER_SLICE_MERGE");
    c->sw.side_stream = !off("SHOULDER_SIDE_STREAM");
    { const char* e = getenv("SHOULDER_UP_INSIDE"); c->sw.up_inside = e && e[0] == '1'; }
    if (const char* e = getenv("SHOULDER_TE_EARLY")) c->sw.te_early = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1);
    c->sw.debug = getenv("SH_DEBUG") != nullptr;
  }
  sh_default_params(&c->params);
  if (hip_stream) c->stream = (hipStream_t)hip_stream;
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SH_ERR_HIP; }
    c->own_stream = true;
  }
  *out = c;
  return SH_OK;
}

static void unet_turn_forget(sh_ctx* c);
static void comm_forget(sh_ctx* c);

void sh_ctx_destroy(sh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->prep.active && c->prep.th.joinable()) c->prep.th.join();
  (void)hipStreamSynchronize(c->stream);
  comm_forget(c);
  unet_turn_forget(c);
  if (c->unet_done_ev) (void)hipEventDestroy(c->unet_done_ev);
  drain_timers(c);
  if (c->stl_counted_ev) (void)hipEventDestroy(c->stl_counted_ev);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
  if (c->side_fork_ev) (void)hipEventDestroy(c->side_fork_ev);
  if (c->side_join_ev) (void)hipEventDestroy(c->side_join_ev);
  if (c->stg.active) (void)hipEventSynchronize(c->stg.ready_ev);
  if (c->stg.h_src) (void)hipHostFree(c->stg.h_src);
  if (c->stg.h_flag) (void)hipHostFree(c->stg.h_flag);
  if (c->stg.h_kept) (void)hipHostFree(c->stg.h_kept);
  if (c->stg.h_koff) (void)hipHostFree(c->stg.h_koff);
  if (c->stg.ready_ev) (void)hipEventDestroy(c->stg.ready_ev);
  if (c->h_kept) (void)hipHostFree(c->h_kept);
  if (c->h_nkept) (void)hipHostFree(c->h_nkept);
  if (c->h_koff) (void)hipHostFree(c->h_koff);
  if (c->obb_done_ev) (void)hipEventDestroy(c->obb_done_ev);
  for (auto& tk : c->tickets) { if (tk.ev) (void)hipEventDestroy(tk.ev); if (tk.h_err) (void)hipHostFree(tk.h_err); }      // (one pinned block: h_ovf and h_fail point into it)
  if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
  for (auto& kv : c->bufs)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& hs : c->hstage) {
    if (hs.hv) { (void)hipHostFree(hs.hv); (void)hipHostFree(hs.nr); (void)hipHostFree(hs.ed); (void)hipHostFree(hs.cnt); }
    if (hs.ev) (void)hipEventDestroy(hs.ev);
  }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int join_prepared(sh_ctx* c);
static void discard_staged(sh_ctx* c);

const char* sh_last_error(const sh_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int sh_set_params(sh_ctx* c, const sh_params* p) {
  if (!c || !p) return SH_ERR_ARG;
  for (double x : {p->groove_cutoff[0], p->groove_cutoff[1], p->canal_cutoff[0], p->canal_cutoff[1]})
    if (!(x >= 0.0 && x <= 1.0)) return fail(c, SH_ERR_ARG, "cut-off fractions must lie in [0, 1]");
  int a, b;
  cutoff_range(SH_NPROX, p->groove_cutoff[0], p->groove_cutoff[1], &a, &b);
  if (b - a != SH_GROOVE_NROWS) return fail(c, SH_ERR_ARG, "groove_cutoff must select 330 proximal rows");
  cutoff_range(SH_NFULL, p->canal_cutoff[0], p->canal_cutoff[1], &a, &b);
  if (b - a < 2 || a < 0 || b > SH_NFULL) return fail(c, SH_ERR_ARG, "canal_cutoff selects fewer than 2 slices");
  // the search window of the groove's local minimum is +-round(deg_window / (360 / 512)) samples of a 512-sample row
  // (bicipital_groove.py:190-229); beyond half a turn the reference's negative indices run off the row (IndexError there)
  if (!(p->groove_deg_window >= 0.0 && p->groove_deg_window <= 180.0)) return fail(c, SH_ERR_ARG, "groove_deg_window must lie in [0, 180] degrees");
  if (p->unet_dtype != SH_UNET_F32 && p->unet_dtype != SH_UNET_BF16 && p->unet_dtype != SH_UNET_F16 && p->unet_dtype != SH_UNET_F32X)
    return fail(c, SH_ERR_ARG, "unet_dtype must be SH_UNET_F32, SH_UNET_F32X, SH_UNET_BF16 or SH_UNET_F16");
  if (p->bone_kind != SH_BONE_HUMERUS && p->bone_kind != SH_BONE_PROXIMAL) return fail(c, SH_ERR_ARG, "bone_kind must be SH_BONE_HUMERUS or SH_BONE_PROXIMAL");
  if (c->prep.active && p->bone_kind != c->params.bone_kind) (void)join_prepared(c);
  c->params = *p;
  c->ovf_none_gen = ~0ull;      // (the plane sets depend on the parameters)
  return SH_OK;
}

int sh_get_params(const sh_ctx* c, sh_params* out) {
  if (!c || !out) return SH_ERR_ARG;
  *out = c->params;
  return SH_OK;
}

int sh_batch_size(const sh_ctx* c) { return c ? c->B : 0; }

// ---- meshes ------------------------------------------------------------------------------------
static int alloc_batch(sh_ctx* c) {
  const int B = c->B;
  int rc;
#define ENS(name, bytes, elem)                                              \
  do {                                                                      \
    if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc;  \
    c->bufs[name].per_mesh = (size_t)(bytes) / (size_t)B;                   \
  } while (0)
  ENS("verts_obb", c->sumV * 3 * 8, 8);
  c->bufs["verts_obb"].per_mesh = 0;        // ragged: indexed through voff
  ENS("verts_csys", c->sumV * 3 * 8, 8);
  c->bufs["verts_csys"].per_mesh = 0;
  c->bufs["voff"].per_mesh = 8;             // a window sees voff[b0 + b] (absolute vertex offsets) as voff[b]
  c->bufs["foff"].per_mesh = 8;
  c->b0 = 0; c->Bwin = B;
  ENS("obb_transform", B * 16 * 8, 8);
  ENS("zb_enc", B * 2 * 8, 8);
  ENS("z_bounds", B * 2 * 8, 8);
  ENS("z_length", B * 8, 8);
  ENS("err", B * 4, 4);
  ENS("neck_z", B * 8, 8);
  ENS("neck_index", B * 4, 4);
  ENS("canal.points_obb", B * SH_CANAL_MAXPTS * 3 * 8, 8);
  ENS("canal.axis_obb", B * 6 * 8, 8);
  ENS("canal.axis_ct", B * 6 * 8, 8);
  ENS("landmarks", (size_t)B * sizeof(sh_landmarks), 1);
  struct S { const char* p; int N; bool ring; };
  const S sets[4] = {{"full", SH_NFULL, false}, {"distal", SH_NDIST, true}, {"prox", SH_NPROX, true}, {"neckc", 1, true}};
  for (const S& s : sets) {
    std::string p = s.p;
    ENS((p + ".zs").c_str(), (size_t)B * s.N * 8, 8);
    ENS((p + ".zeff").c_str(), (size_t)B * s.N * 8, 8);
    ENS((p + ".seg_count").c_str(), (size_t)B * s.N * 4, 4);
    ENS((p + ".segs").c_str(), (size_t)B * s.N * SH_MAXSEG * sizeof(Seg), 1);
    ENS((p + ".centroids").c_str(), (size_t)B * s.N * 2 * 8, 8);
    ENS((p + ".areas").c_str(), (size_t)B * s.N * 8, 8);
    ENS((p + ".nloops").c_str(), (size_t)B * s.N * 4, 4);
    ENS((p + ".ring_n").c_str(), (size_t)B * s.N * 4, 4);
    if (s.ring) ENS((p + ".ring").c_str(), (size_t)B * s.N * (SH_MAXSEG + 1) * 2 * 8, 8);
  }
  ENS("slices.nlarge", 64, 4);
  c->bufs["slices.nlarge"].per_mesh = 0;
  ENS("prox.ixy", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
  ENS("prox.itr_start", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
  ENS("prox.itr_centered_start", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
  // groove
  ENS("groove.xraw", (size_t)B * SH_GSLOTS * 9 * 8, 8);
  ENS("groove.xs", (size_t)B * SH_GSLOTS * 9 * 8, 8);
  ENS("groove.ptheta", (size_t)B * SH_GSLOTS * 8, 8);
  ENS("groove.npk", (size_t)B * SH_GROOVE_NROWS * 4, 4);
  ENS("groove.r0", (size_t)B * SH_GROOVE_NROWS * SH_MPROX * 8, 8);
  ENS("groove.stats", (size_t)B * 18 * 8, 8);
  ENS("groove.proba", (size_t)B * SH_GSLOTS * 4, 4);
  ENS("groove.slots", (size_t)B * SH_GSLOTS * 4, 4);
  ENS("groove.nslot", (size_t)B * 4, 4);
  ENS("groove.bg_theta", (size_t)B * 8, 8);
  ENS("groove.local_idx", (size_t)B * SH_GROOVE_NROWS * 4, 4);
  ENS("groove.points_obb", (size_t)B * SH_GROOVE_NROWS * 3 * 8, 8);
  ENS("groove.points_ct", (size_t)B * SH_GROOVE_NROWS * 3 * 8, 8);
  ENS("groove.axis_ct", (size_t)B * 6 * 8, 8);
  // anatomic neck
  ENS("anp.raw", (size_t)B * SH_IMG * 8, 8);
  ENS("anp.t01", (size_t)B * SH_ANP_ROWS * 2 * 8, 8);
  ENS("anp.roll", (size_t)B * SH_ANP_ROWS * 4, 4);
  ENS("anp.maskbits", (size_t)B * SH_ANP_ROWS * (SH_MPROX / 64) * 8, 8);
  ENS("anp.mm_enc", (size_t)B * 2 * 8, 8);
  ENS("metrics.partial", (size_t)B * SH_SPH_PARTS * 14 * 8, 8);
  ENS("anp.image", (size_t)B * SH_IMG * 4, 4);
  ENS("anp.logits", (size_t)B * SH_IMG * 4, 4);
  ENS("anp.points_obb", (size_t)B * SH_ANP_CAP * 3 * 8, 8);
  ENS("anp.counts", (size_t)B * 2 * 4, 4);
  ENS("anp.rowcnt", (size_t)B * SH_ANP_ROWS * 2 * 4, 4);
  ENS("anp.ray_t", (size_t)B * 4 * 8, 8);
  ENS("anp.plane", (size_t)B * 6 * 8, 8);
  ENS("anp.axes_obb", (size_t)B * 12 * 8, 8);
  // trans-epicondylar
  ENS("te.rects", (size_t)B * SH_TE_NROWS * 7 * 8, 8);
  ENS("te.axis_ct", (size_t)B * 6 * 8, 8);
  ENS("te.ends_ct", (size_t)B * 6 * 8, 8);
  ENS("te.row", (size_t)B * 4, 4);
  ENS("flipped", (size_t)B * 4, 4);
  HIPCHK(c, hipMemsetAsync(buf<int>(c, "flipped"), 0, (size_t)B * 4, c->stream));
  // oriented bounding box
  ENS("hull.hv", (size_t)B * c->hcap.v * 3 * 8, 8);
  ENS("hull.normals", (size_t)B * c->hcap.f * 3 * 8, 8);
  ENS("hull.edges", (size_t)B * c->hcap.e * 4 * 4, 4);
  ENS("hull.nv", (size_t)B * 4, 4);
  ENS("hull.nf", (size_t)B * 4, 4);
  ENS("hull.ne", (size_t)B * 4, 4);
  ENS("hullpre.ext", (size_t)B * SH_HP_NDIR * 4, 4);
  ENS("hullpre.planes", (size_t)B * SH_HP_MAXPL * 4 * 8, 8);
  ENS("hullpre.npl", (size_t)B * 4, 4);
  ENS("hullpre.nkept", (size_t)B * 4, 4);
  ENS("hullpre.pval", (size_t)B * SH_HP_PARTS * SH_HP_NDIR * 8, 8);
  ENS("hullpre.pidx", (size_t)B * SH_HP_PARTS * SH_HP_NDIR * 4, 4);
  ENS("hullpre.pcnt", (size_t)B * SH_HP_PARTS * 4, 4);
  ENS("hullpre.poff", (size_t)B * SH_HP_PARTS * 8, 8);
  ENS("hullpre.koff", (size_t)(B + 1) * 8, 8);
  ENS("hullpre.kept", (size_t)c->sumV * 12, 4);
  if (c->h_kept_cap < c->sumV) {
    if (c->h_kept) (void)hipHostFree(c->h_kept);
    c->h_kept = nullptr; c->h_kept_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_kept, (size_t)c->sumV * 12));
    c->h_kept_cap = c->sumV;
  }
  if (c->h_nkept_cap < B) {
    if (c->h_nkept) (void)hipHostFree(c->h_nkept);
    if (c->h_koff) (void)hipHostFree(c->h_koff);
    c->h_nkept = nullptr; c->h_koff = nullptr; c->h_nkept_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_nkept, (size_t)B * 4));
    HIPCHK(c, hipHostMalloc((void**)&c->h_koff, (size_t)(B + 1) * 8));
    c->h_nkept_cap = B;
  }
  ENS("obb.cand_vol", (size_t)B * c->hcap.f * 8, 8);
  ENS("obb.cand_edge", (size_t)B * c->hcap.f * 4, 4);
  ENS("obb.best_enc", (size_t)B * 8, 8);
  ENS("obb.lbmin_enc", (size_t)B * 8, 8);
  ENS("obb.area2", (size_t)B * c->hcap.f * 8, 8);
  ENS("obb.lb", (size_t)B * c->hcap.f * 8, 8);
  ENS("obb.dir_list", (size_t)B * c->hcap.f * 4, 4);
  ENS("obb.dir_count", (size_t)B * 4, 4);
  ENS("obb.seeded", (size_t)B * c->hcap.f, 1);
  ENS("obb.T_pre", (size_t)B * 16 * 8, 8);
  ENS("obb.zb_pre", (size_t)B * 2 * 8, 8);
  ENS("obb.endpts", (size_t)B * 2 * c->end_cap * 2 * 8, 8);
  ENS("obb.endcnt", (size_t)B * 2 * 4, 4);
  ENS("obb.resid", (size_t)B * 2 * 8, 8);
#undef ENS
  c->obb_injected = false;
  return SH_OK;
}

// Scratch of the device hull (k_hull.h), ~1.6 MB per humerus, allocated on the first run that uses it.
static int alloc_hulld(sh_ctx* c) {
  const int B = c->B;
  int rc;
#define ENS(name, bytes, elem)                                              \
  do {                                                                      \
    if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc;  \
    c->bufs[name].per_mesh = (size_t)(bytes) / (size_t)B;                   \
  } while (0)
  ENS("hulld.fv", (size_t)B * HD_SLOTS * 3 * 4, 4);
  ENS("hulld.vis", (size_t)B * HD_KC * HD_VMAX * 4, 4);
  ENS("hulld.ev", (size_t)B * HD_KC * 3 * HD_VMAX * 2 * 4, 4);
  ENS("hulld.hor", (size_t)B * HD_KC * (HD_VMAX + 2) * 2 * 4, 4);
  ENS("hulld.newslot", (size_t)B * HD_SLOTS * 4, 4);
  ENS("hulld.freestack", (size_t)B * HD_SLOTS * 4, 4);
  ENS("hulld.tkeys", (size_t)B * HD_TBL * 8, 8);
  ENS("hulld.tvals", (size_t)B * HD_TBL * 4, 4);
  ENS("hulld.fail", (size_t)B * 4, 4);
  ENS("hulld.rounds", (size_t)B * 4, 4);
  ENS("hulld.skip", (size_t)B * 4, 4);
#undef ENS
  if (c->skip_gen != c->batch_gen) {      // a new batch: the device hull takes every humerus again
    HIPCHK(c, hipMemsetAsync(c->bufs["hulld.skip"].p, 0, (size_t)B * 4, c->stream));
    c->skip_gen = c->batch_gen;
    c->skip_nfmax = 0;
  }
  return SH_OK;
}

// Extra buffers of the SH_BONE_PROXIMAL path (allocated on first use): the ProxObb area scan and the large Gram matrix
// of the neck change point.
static int alloc_prox(sh_ctx* c) {
  const int B = c->B;
  int rc;
#define ENS(name, bytes, elem)                                              \
  do {                                                                      \
    if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc;  \
    c->bufs[name].per_mesh = (size_t)(bytes) / (size_t)B;                   \
  } while (0)
  const int N = SH_NPSCAN;
  ENS("pobb.zs", (size_t)B * N * 8, 8);
  ENS("pobb.zeff", (size_t)B * N * 8, 8);
  ENS("pobb.seg_count", (size_t)B * N * 4, 4);
  ENS("pobb.segs", (size_t)B * N * SH_MAXSEG * sizeof(Seg), 1);
  ENS("pobb.centroids", (size_t)B * N * 2 * 8, 8);
  ENS("pobb.areas", (size_t)B * N * 8, 8);
  ENS("pobb.nloops", (size_t)B * N * 4, 4);
  ENS("pobb.ring_n", (size_t)B * N * 4, 4);
  ENS("pobb.area_total", (size_t)B * N * 8, 8);
  ENS("pobb.cutoff", (size_t)B * 2 * 8, 8);
  ENS("pobb.cutoff_idx", (size_t)B * 2 * 4, 4);
  ENS("neck.gram", (size_t)B * SH_NFULL * SH_NFULL * 8, 8);
#undef ENS
  return SH_OK;
}

int sh_upload_meshes(sh_ctx* c, const float* verts, const int32_t* faces, const int64_t* v_off, const int64_t* f_off, int B) {
  if (!c || !verts || !faces || !v_off || !f_off || B <= 0) return fail(c, SH_ERR_ARG, "sh_upload_meshes: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  // Validate first, into locals: a rejected upload leaves the resident batch (B, offsets, device buffers) untouched.
  if (v_off[0] != 0 || f_off[0] != 0) return fail(c, SH_ERR_ARG, "sh_upload_meshes: offsets must start at 0");
  long long maxV = 0, maxF = 0;
  for (int b = 0; b < B; ++b) {
    long long nv = v_off[b + 1] - v_off[b], nf = f_off[b + 1] - f_off[b];
    if (nv < 4 || nf < 4) return fail(c, SH_ERR_ARG, "sh_upload_meshes: a mesh has fewer than 4 vertices/faces");
    if (nv > 0x7fffffffLL / 3 || nf > 0x7fffffffLL / 3) return fail(c, SH_ERR_ARG, "sh_upload_meshes: a mesh is too large");
    maxV = std::max(maxV, nv); maxF = std::max(maxF, nf);
    for (long long i = 3 * f_off[b]; i < 3 * f_off[b + 1]; ++i)
      if (faces[i] < 0 || faces[i] >= nv) return fail(c, SH_ERR_ARG, "sh_upload_meshes: face index out of range");
  }
  const long long sumV = v_off[B], sumF = f_off[B];
  for (long long i = 0; i < 3 * sumV; ++i)
    if (!std::isfinite(verts[i])) return fail(c, SH_ERR_ARG, "sh_upload_meshes: NaN / infinite vertex coordinate");
  discard_staged(c); (void)join_prepared(c); ++c->batch_gen;      // hulls prepared for the previous batch are void
  c->B = 0;                                     // (a HIP / allocation failure below leaves "no meshes uploaded", never a half-committed batch)
  c->h_voff.assign(v_off, v_off + B + 1);
  c->h_foff.assign(f_off, f_off + B + 1);
  c->sumV = sumV; c->sumF = sumF; c->maxV = maxV; c->maxF = maxF;
  c->h_verts.assign(verts, verts + 3 * c->sumV);
  c->h_verts_valid = true;
  int rc;
  if ((rc = ensure(c, "verts", c->sumV * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "faces", c->sumF * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "voff", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "foff", (B + 1) * 8, 8)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<float>(c, "verts"), verts, c->sumV * 3 * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<int>(c, "faces"), faces, c->sumF * 3 * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "voff"), c->h_voff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "foff"), c->h_foff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = B;
  if ((rc = alloc_batch(c)) != SH_OK) c->B = 0;
  return rc;
}

// Binary STL files -> merged meshes, on the device (k_stl.h; replaces `trimesh.load_mesh(stl)` of mesh.py:22-27 incl. the
// vertex merge).  The host only checks the 84-byte headers and sums sizes.
int sh_upload_stl(sh_ctx* c, const void* const* files, const size_t* nbytes, int B, int64_t* v_off_out, int64_t* f_off_out) {
  if (!c || !files || !nbytes || B <= 0) return fail(c, SH_ERR_ARG, "sh_upload_stl: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  discard_staged(c);      // (a staged STL batch works in the same stl.* scratch)
  std::vector<long long> file_off(B + 1, 0), coff(B + 1, 0);
  long long maxc = 0;
  for (int b = 0; b < B; ++b) {
    if (!files[b] || nbytes[b] < 84) return fail(c, SH_ERR_ARG, "sh_upload_stl: a file is too short for a binary STL");
    uint32_t nt;
    memcpy(&nt, (const char*)files[b] + 80, 4);
    if (nbytes[b] != 84 + 50ull * nt) return fail(c, SH_ERR_ARG, "sh_upload_stl: not a binary STL (size does not match the triangle count)");
    if (nt < 4 || nt > 0x7fffffffu / 3) return fail(c, SH_ERR_ARG, "sh_upload_stl: a mesh has fewer than 4 (or too many) triangles");
    file_off[b + 1] = file_off[b] + (long long)((nbytes[b] + 3) & ~(size_t)3);
    coff[b + 1] = coff[b] + 3ll * nt;
    maxc = std::max(maxc, 3ll * nt);
  }
  int tsize = 1024;
  while (tsize < 2 * maxc) tsize <<= 1;
  const long long sumC = coff[B];
  int rc;
  if ((rc = ensure(c, "stl.raw", (size_t)file_off[B], 1)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.file_off", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.coff", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.corners", (size_t)sumC * 12, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.table", (size_t)B * tsize * 8, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.slot", (size_t)sumC * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.vid", (size_t)sumC * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.fpos", (size_t)(sumC / 3) * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.counts", (size_t)B * 8, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "stl.bsum", stl_rank_scratch_ints(B, maxc) * 4, 4)) != SH_OK) return rc;
  c->bufs["stl.bsum"].per_mesh = 0;
  if ((rc = ensure(c, "stl.nonfinite", (size_t)B * 4, 4)) != SH_OK) return rc;
  c->bufs["stl.nonfinite"].per_mesh = 0;
  HIPCHK(c, hipMemsetAsync(c->bufs["stl.nonfinite"].p, 0, (size_t)B * 4, c->stream));
  for (const char* nm : {"stl.raw", "stl.file_off", "stl.coff", "stl.corners", "stl.table", "stl.slot", "stl.vid", "stl.fpos", "stl.counts"}) c->bufs[nm].per_mesh = 0;
  c->b0 = 0;
  unsigned char* raw = buf<unsigned char>(c, "stl.raw");
  for (int b = 0; b < B; ++b) HIPCHK(c, hipMemcpyAsync(raw + file_off[b], files[b], nbytes[b], hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "stl.file_off"), file_off.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "stl.coff"), coff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  const dim3 gc((unsigned)std::min<long long>((maxc + 255) / 256, 1024), (unsigned)B);
  LAUNCH(c, "k_stl_corners", k_stl_corners, gc, dim3(256), raw, buf<long long>(c, "stl.file_off"), buf<long long>(c, "stl.coff"), buf<float>(c, "stl.corners"), (int*)c->bufs["stl.nonfinite"].p);
  LAUNCH(c, "k_stl_table_init", k_stl_table_init, dim3(1024), dim3(256), buf<int2>(c, "stl.table"), (size_t)B * tsize);
  LAUNCH(c, "k_stl_hash", k_stl_hash, gc, dim3(256), buf<float>(c, "stl.corners"), buf<long long>(c, "stl.coff"), buf<int2>(c, "stl.table"), tsize, buf<int>(c, "stl.slot"));
  stl_rank_launch(c->stream, B, maxc, buf<long long>(c, "stl.coff"), buf<int2>(c, "stl.table"), tsize, buf<int>(c, "stl.slot"), buf<int>(c, "stl.vid"), buf<int>(c, "stl.fpos"),
                  buf<int>(c, "stl.counts"), buf<int>(c, "stl.bsum"));
  HIPCHK(c, hipGetLastError());
  std::vector<int> counts(2 * B), nonfin(B);
  HIPCHK(c, hipMemcpyAsync(counts.data(), buf<int>(c, "stl.counts"), (size_t)B * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(nonfin.data(), c->bufs["stl.nonfinite"].p, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // Validate into locals: a rejected file set leaves the resident batch (B, offsets, "verts" / "faces") untouched -- only
  // the stl.* scratch buffers were written so far.
  std::vector<long long> n_voff(B + 1, 0), n_foff(B + 1, 0);
  long long maxV = 0, maxF = 0;
  for (int b = 0; b < B; ++b) {
    if (nonfin[b]) return fail(c, SH_ERR_ARG, "sh_upload_stl: a file holds NaN / infinite coordinates");
    if (counts[2 * b] < 4 || counts[2 * b + 1] < 4) return fail(c, SH_ERR_ARG, "sh_upload_stl: a mesh has fewer than 4 vertices/faces after merging");
    n_voff[b + 1] = n_voff[b] + counts[2 * b];
    n_foff[b + 1] = n_foff[b] + counts[2 * b + 1];
    maxV = std::max<long long>(maxV, counts[2 * b]); maxF = std::max<long long>(maxF, counts[2 * b + 1]);
  }
  discard_staged(c); (void)join_prepared(c); ++c->batch_gen;
  c->B = 0;                        // (a HIP / allocation failure below leaves "no meshes uploaded")
  c->h_voff.swap(n_voff); c->h_foff.swap(n_foff);
  c->maxV = maxV; c->maxF = maxF;
  c->sumV = c->h_voff[B]; c->sumF = c->h_foff[B];
  c->h_verts_valid = false;      // the hull stage downloads the merged vertices (as for a device-generated batch)
  if ((rc = ensure(c, "verts", c->sumV * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "faces", c->sumF * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "voff", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "foff", (B + 1) * 8, 8)) != SH_OK) return rc;
  c->bufs["verts"].per_mesh = 0; c->bufs["faces"].per_mesh = 0;
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "voff"), c->h_voff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "foff"), c->h_foff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_stl_emit", k_stl_emit, gc, dim3(256), buf<float>(c, "stl.corners"), buf<long long>(c, "stl.coff"), buf<int2>(c, "stl.table"), tsize, buf<int>(c, "stl.slot"),
         buf<int>(c, "stl.vid"), buf<int>(c, "stl.fpos"), buf<long long>(c, "voff"), buf<long long>(c, "foff"), buf<float>(c, "verts"), buf<int>(c, "faces"));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (v_off_out) for (int b = 0; b <= B; ++b) v_off_out[b] = c->h_voff[b];
  if (f_off_out) for (int b = 0; b <= B; ++b) f_off_out[b] = c->h_foff[b];
  c->B = B;
  if ((rc = alloc_batch(c)) != SH_OK) c->B = 0;
  return rc;
}

int sh_synth_batch(sh_ctx* c, const double* T, int B) {
  if (!c || !T || B <= 0) return fail(c, SH_ERR_ARG, "sh_synth_batch: bad argument");
  if (c->B < 1) return fail(c, SH_ERR_STATE, "sh_synth_batch: upload a template mesh first");
  HIPCHK(c, hipSetDevice(c->device));
  discard_staged(c); (void)join_prepared(c); ++c->batch_gen;
  const long long V = c->h_voff[1] - c->h_voff[0], F = c->h_foff[1] - c->h_foff[0];
  // keep the template aside
  int rc;
  if ((rc = ensure(c, "tmpl_verts", V * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "tmpl_faces", F * 3 * 4, 4)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<float>(c, "tmpl_verts"), buf<float>(c, "verts") + 3 * c->h_voff[0], V * 3 * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<int>(c, "tmpl_faces"), buf<int>(c, "faces") + 3 * c->h_foff[0], F * 3 * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = 0;                        // (a failure below leaves "no meshes uploaded", never a half-committed batch)
  c->h_verts_valid = false;
  c->sumV = V * B; c->sumF = F * B; c->maxV = V; c->maxF = F;
  c->h_voff.resize(B + 1); c->h_foff.resize(B + 1);
  for (int b = 0; b <= B; ++b) { c->h_voff[b] = V * b; c->h_foff[b] = F * b; }
  if ((rc = ensure(c, "verts", c->sumV * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "faces", c->sumF * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "voff", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "foff", (B + 1) * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "synth_T", B * 16 * 8, 8)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<double>(c, "synth_T"), T, B * 16 * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "voff"), c->h_voff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "foff"), c->h_foff.data(), (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  dim3 grid((unsigned)((V + 255) / 256), (unsigned)B);
  LAUNCH(c, "k_synth_batch", k_synth_batch, grid, dim3(256), buf<float>(c, "tmpl_verts"), buf<int>(c, "tmpl_faces"),
         (long long)V, (long long)F, buf<double>(c, "synth_T"), buf<float>(c, "verts"), buf<int>(c, "faces"));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = B;
  if ((rc = alloc_batch(c)) != SH_OK) c->B = 0;
  return rc;
}

// ---- named buffers -----------------------------------------------------------------------------
int sh_buffer_info(sh_ctx* c, const char* name, size_t* nbytes, int* elem) {
  if (!c || !name) return SH_ERR_ARG;
  auto it = c->bufs.find(name);
  if (it == c->bufs.end()) return fail(c, SH_ERR_ARG, std::string("no buffer named ") + name);
  if (nbytes) *nbytes = it->second.bytes;
  if (elem) *elem = it->second.elem;
  return SH_OK;
}

int sh_buffer_device(sh_ctx* c, const char* name, void** dev_ptr, size_t* nbytes) {
  if (!c || !name || !dev_ptr) return SH_ERR_ARG;
  auto it = c->bufs.find(name);
  if (it == c->bufs.end() || !it->second.p) return fail(c, SH_ERR_ARG, std::string("no buffer named ") + name);
  *dev_ptr = it->second.p;
  if (nbytes) *nbytes = it->second.bytes;
  if (std::string(name) == "params") { c->packed_kind = -1; c->packed_x3 = false; c->packed_rfc = false; }      // (the caller may write it)
  c->ovf_none_gen = ~0ull;      // (... or a frame / an intermediate that moves the planes: the overflow tier runs again)
  return SH_OK;
}

int sh_fetch(sh_ctx* c, const char* name, void* host, size_t nbytes) {
  if (!c || !name || !host) return SH_ERR_ARG;
  auto it = c->bufs.find(name);
  if (it == c->bufs.end()) return fail(c, SH_ERR_ARG, std::string("no buffer named ") + name);
  if (nbytes > it->second.bytes) return fail(c, SH_ERR_ARG, std::string("sh_fetch: size exceeds buffer ") + name);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(host, it->second.p, nbytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

int sh_store(sh_ctx* c, const char* name, const void* host, size_t nbytes) {
  if (!c || !name || !host) return SH_ERR_ARG;
  auto it = c->bufs.find(name);
  if (it == c->bufs.end()) return fail(c, SH_ERR_ARG, std::string("no buffer named ") + name);
  if (nbytes > it->second.bytes) return fail(c, SH_ERR_ARG, std::string("sh_store: size exceeds buffer ") + name);
  HIPCHK(c, hipSetDevice(c->device));
  if (std::string(name) == "verts") { discard_staged(c); (void)join_prepared(c); ++c->batch_gen; c->h_verts_valid = false; }
  c->ovf_none_gen = ~0ull;      // (an injected frame or intermediate moves the planes: the overflow tier runs again)
  if (std::string(name) == "params") { c->packed_kind = -1; c->packed_x3 = false; c->packed_rfc = false; }
  HIPCHK(c, hipMemcpyAsync(it->second.p, host, nbytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (std::string(name) == "obb_transform") c->obb_injected = true;
  return SH_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int sh_enable_timing(sh_ctx* c, int on) {
  if (!c) return SH_ERR_ARG;
  (void)hipStreamSynchronize(c->stream);
  drain_timers(c);
  c->timing = on < 0 ? 0 : (on > 2 ? 1 : on);
  return SH_OK;
}

int sh_kernel_time_ms(sh_ctx* c, const char* kernel, double* avg_ms, int* launches) {
  if (!c) return SH_ERR_ARG;
  (void)hipStreamSynchronize(c->stream);
  drain_timers(c);
  if (!kernel) { c->timers.clear(); return SH_OK; }
  auto it = c->timers.find(kernel);
  if (it == c->timers.end() || it->second.n == 0) { if (avg_ms) *avg_ms = 0; if (launches) *launches = 0; return SH_OK; }
  if (avg_ms) *avg_ms = it->second.ms / it->second.n;
  if (launches) *launches = it->second.n;
  return SH_OK;
}

// ---- affine --------------------------------------------------------------------------------------
int sh_affine_apply(sh_ctx* c, const double* T, const void* dev_in, void* dev_out, const int64_t* off, int B) {
  if (!c || !T || !dev_in || !dev_out || !off || B <= 0) return fail(c, SH_ERR_ARG, "sh_affine_apply: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, "aff_T", B * 16 * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "aff_off", (B + 1) * 8, 8)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<double>(c, "aff_T"), T, B * 16 * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "aff_off"), off, (B + 1) * 8, hipMemcpyHostToDevice, c->stream));
  long long mx = 0;
  for (int b = 0; b < B; ++b) mx = std::max<long long>(mx, off[b + 1] - off[b]);
  dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((mx + 255) / 256, 1024)), (unsigned)B);
  LAUNCH(c, "k_affine_f64", k_affine_f64, grid, dim3(256), buf<double>(c, "aff_T"), (const double*)dev_in, (double*)dev_out,
         buf<long long>(c, "aff_off"));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

int sh_transform_points(sh_ctx* c, const double* T, const double* in, int n, double* out) {
  if (!c || !T || !in || !out || n < 0) return fail(c, SH_ERR_ARG, "sh_transform_points: bad argument");
  if (n == 0) return SH_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, "tp_T", 16 * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "tp_io", (size_t)n * 3 * 8 * 2, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "tp_off", 2 * 8, 8)) != SH_OK) return rc;
  long long off[2] = {0, n};
  double* io = buf<double>(c, "tp_io");
  HIPCHK(c, hipMemcpyAsync(buf<double>(c, "tp_T"), T, 16 * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<long long>(c, "tp_off"), off, 16, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(io, in, (size_t)n * 3 * 8, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_affine_f64", k_affine_f64, dim3((unsigned)std::min(1024, (n + 255) / 256), 1), dim3(256), buf<double>(c, "tp_T"), io, io + (size_t)n * 3,
         buf<long long>(c, "tp_off"));
  HIPCHK(c, hipMemcpyAsync(out, io + (size_t)n * 3, (size_t)n * 3 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

// The closed largest loop of plane k of slice set `set` ("full" has none; "distal", "prox", "neckc") of humerus b after a run:
// n + 1 points (x, y) in the box frame, CCW, canonical start, first = last.  From the fixed slot range or, for a plane with more
// crossings than slots, from the overflow pool (k_ovf.h).  out == NULL or cap too small: *n_out = n + 1, nothing copied.
int sh_ring(sh_ctx* c, const char* set, int b, int k, double* out, int cap, int* n_out) {
  if (!c || !set || !n_out) return fail(c, SH_ERR_ARG, "sh_ring: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const std::string p = set;
  auto itn = c->bufs.find(p + ".ring_n");
  auto itr = c->bufs.find(p + ".ring");
  if (itn == c->bufs.end() || itr == c->bufs.end() || !itr->second.p) return fail(c, SH_ERR_ARG, "sh_ring: no such slice set with rings");
  const int N = (int)(itn->second.per_mesh / 4);
  if (b < 0 || b >= c->B || k < 0 || k >= N) return fail(c, SH_ERR_ARG, "sh_ring: index out of range");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t pl = (size_t)b * N + k;
  int n = 0;
  HIPCHK(c, hipMemcpy(&n, (const int*)itn->second.p + pl, 4, hipMemcpyDeviceToHost));
  *n_out = n + 1;
  if (!out || cap < n + 1) return SH_OK;
  long long roff = -1;
  auto ito = c->bufs.find(p + ".ovf_roff");
  if (ito != c->bufs.end() && ito->second.p) HIPCHK(c, hipMemcpy(&roff, (const long long*)ito->second.p + pl, 8, hipMemcpyDeviceToHost));
  const double* src = roff >= 0 ? (const double*)c->bufs["ovf.ring"].p + 2 * roff : (const double*)itr->second.p + pl * (SH_MAXSEG + 1) * 2;
  HIPCHK(c, hipMemcpy(out, src, (size_t)(n + 1) * 16, hipMemcpyDeviceToHost));
  return SH_OK;
}

int sh_section_plane(sh_ctx* c, int b, const double* origin, const double* normal, double* out_pts, int cap, int* n_out) {
  if (!c || !origin || !normal || !out_pts || !n_out || cap <= 0 || b < 0 || b >= c->B) return fail(c, SH_ERR_ARG, "sh_section_plane: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  double pl[6] = {origin[0], origin[1], origin[2], normal[0], normal[1], normal[2]};
  double nn = std::sqrt(pl[3] * pl[3] + pl[4] * pl[4] + pl[5] * pl[5]);
  if (!(nn > 0)) return fail(c, SH_ERR_ARG, "sh_section_plane: zero normal");
  for (int k = 3; k < 6; ++k) pl[k] /= nn;
  int rc;
  if ((rc = ensure(c, "sp_plane", 6 * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "sp_out", (size_t)cap * 3 * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "sp_cnt", 4, 4)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<double>(c, "sp_plane"), pl, 48, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(buf<int>(c, "sp_cnt"), 0, 4, c->stream));
  long long nf = c->h_foff[b + 1] - c->h_foff[b];
  LAUNCH(c, "k_section_points", k_section_points, dim3((unsigned)std::min<long long>((nf + 255) / 256, 1024)), dim3(256),
         buf<float>(c, "verts") + 3 * c->h_voff[b], buf<int>(c, "faces") + 3 * c->h_foff[b], nf, buf<double>(c, "sp_plane"), buf<double>(c, "sp_out"), cap,
         buf<int>(c, "sp_cnt"));
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, buf<int>(c, "sp_cnt"), 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  if (n > cap) return fail(c, SH_ERR_CAPACITY, "sh_section_plane: more crossing points than capacity");
  HIPCHK(c, hipMemcpy(out_pts, buf<double>(c, "sp_out"), (size_t)n * 3 * 8, hipMemcpyDeviceToHost));
  return SH_OK;
}

// Cut one mesh (host arrays, any coordinate system) with P planes and keep the side each normal points to
// (`Trimesh.slice_plane`, arthroplasty.py:80-87).  out_verts == nullptr: count only (counts[].n_verts = upper bound).
int sh_slice_mesh_planes(sh_ctx* c, const double* verts, int nv, const int32_t* faces, int nf, const double* origins, const double* normals, int P,
                         double* out_verts, int cap_v, int32_t* out_faces, int cap_f, int32_t* out_edges, int cap_e, int32_t* counts) {
  if (!c || !verts || !faces || !origins || !normals || !counts || nv < 1 || nf < 1 || P < 1 || P > 4096)
    return fail(c, SH_ERR_ARG, "sh_slice_mesh_planes: bad argument");
  if (out_verts && (!out_faces || cap_v < 1 || cap_f < 1 || (out_edges && cap_e < 1))) return fail(c, SH_ERR_ARG, "sh_slice_mesh_planes: bad output arguments");
  for (int i = 0; i < 3 * nf; ++i)
    if (faces[i] < 0 || faces[i] >= nv) return fail(c, SH_ERR_ARG, "sh_slice_mesh_planes: face index out of range");
  if ((long long)P * ((long long)nv + 2LL * nf) > (1LL << 30)) return fail(c, SH_ERR_CAPACITY, "sh_slice_mesh_planes: planes x mesh too large for one call");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<double> pl(6 * (size_t)P);
  for (int p = 0; p < P; ++p) {
    const double* n = normals + 3 * p;
    if (!((n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 0)) return fail(c, SH_ERR_ARG, "sh_slice_mesh_planes: zero normal");
    for (int k = 0; k < 3; ++k) { pl[6 * p + k] = origins[3 * p + k]; pl[6 * p + 3 + k] = n[k]; }      // the normal is used as given (trimesh does not normalise it)
  }
  void *d_v, *d_f, *d_pl, *d_sign, *d_cls, *d_fpos, *d_cnt;
  int rc;
  if ((rc = ensure(c, "clip.verts", (size_t)nv * 24, 8, &d_v)) || (rc = ensure(c, "clip.faces", (size_t)nf * 12, 4, &d_f)) ||
      (rc = ensure(c, "clip.planes", (size_t)P * 48, 8, &d_pl)) || (rc = ensure(c, "clip.sign", (size_t)P * nv, 1, &d_sign)) ||
      (rc = ensure(c, "clip.cls", (size_t)P * nf, 1, &d_cls)) || (rc = ensure(c, "clip.fpos", (size_t)P * nf * 4, 4, &d_fpos)) ||
      (rc = ensure(c, "clip.counts", (size_t)P * sizeof(ClipCounts), 4, &d_cnt)))
    return rc;
  HIPCHK(c, hipMemcpyAsync(d_v, verts, (size_t)nv * 24, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_f, faces, (size_t)nf * 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_pl, pl.data(), (size_t)P * 48, hipMemcpyHostToDevice, c->stream));
  const unsigned gv = (unsigned)std::min(256, (nv + 255) / 256), gf = (unsigned)std::min(256, (nf + 255) / 256);
  LAUNCH(c, "k_clip_sign", k_clip_sign, dim3(gv, P), dim3(256), (const double*)d_v, nv, (const double*)d_pl, (signed char*)d_sign);
  LAUNCH(c, "k_clip_class", k_clip_class, dim3(P), dim3(SH_STL_SCAN_THREADS), (const double*)d_v, (const int*)d_f, nf, nv, (const double*)d_pl,
         (const signed char*)d_sign, (unsigned char*)d_cls, (int*)d_fpos, (ClipCounts*)d_cnt);
  std::vector<ClipCounts> cn(P);
  HIPCHK(c, hipMemcpyAsync(cn.data(), d_cnt, (size_t)P * sizeof(ClipCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  auto report = [&](bool upper) {
    for (int p = 0; p < P; ++p) { counts[3 * p] = upper ? cn[p].n_pre : cn[p].n_verts; counts[3 * p + 1] = cn[p].n_faces; counts[3 * p + 2] = cn[p].n_edges; }
  };
  if (!out_verts) { report(true); return SH_OK; }
  // per-plane offsets of the pre-merge arrays
  std::vector<long long> off(4 * (size_t)(P + 1) + (P + 1), 0);
  long long *new_off = off.data(), *face_off = new_off + (P + 1), *edge_off = face_off + (P + 1), *pre_off = edge_off + (P + 1), *tab_off = pre_off + (P + 1);
  for (int p = 0; p < P; ++p) {
    if (cn[p].n_faces > cap_f || (out_edges && cn[p].n_edges > cap_e)) { report(true); return fail(c, SH_ERR_CAPACITY, "sh_slice_mesh_planes: output capacity too small (counts hold the sizes needed)"); }
    new_off[p + 1] = new_off[p] + 2LL * (cn[p].n_quad + cn[p].n_tri);
    face_off[p + 1] = face_off[p] + cn[p].n_faces;
    edge_off[p + 1] = edge_off[p] + cn[p].n_edges;
    pre_off[p + 1] = pre_off[p] + cn[p].n_pre;
    long long ts = 1024;
    while (ts < 2LL * cn[p].n_pre) ts <<= 1;
    tab_off[p + 1] = tab_off[p] + ts;
  }
  void *d_off, *d_np, *d_pf, *d_pe, *d_ref, *d_keys, *d_tab, *d_slot, *d_vid, *d_ov, *d_of, *d_oe = nullptr;
  if ((rc = ensure(c, "clip.off", off.size() * 8, 8, &d_off)) || (rc = ensure(c, "clip.new_pts", (size_t)std::max(1LL, new_off[P]) * 24, 8, &d_np)) ||
      (rc = ensure(c, "clip.pre_faces", (size_t)std::max(1LL, face_off[P]) * 12, 4, &d_pf)) || (rc = ensure(c, "clip.pre_edges", (size_t)std::max(1LL, edge_off[P]) * 8, 4, &d_pe)) ||
      (rc = ensure(c, "clip.referenced", (size_t)pre_off[P], 1, &d_ref)) || (rc = ensure(c, "clip.keys", (size_t)pre_off[P] * 24, 8, &d_keys)) ||
      (rc = ensure(c, "clip.table", (size_t)tab_off[P] * 8, 8, &d_tab)) || (rc = ensure(c, "clip.slot", (size_t)pre_off[P] * 4, 4, &d_slot)) ||
      (rc = ensure(c, "clip.vid", (size_t)pre_off[P] * 4, 4, &d_vid)) || (rc = ensure(c, "clip.out_verts", (size_t)P * cap_v * 24, 8, &d_ov)) ||
      (rc = ensure(c, "clip.out_faces", (size_t)P * cap_f * 12, 4, &d_of)))
    return rc;
  if (out_edges && (rc = ensure(c, "clip.out_edges", (size_t)P * cap_e * 8, 4, &d_oe))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream));
  const long long *g_new = (const long long*)d_off, *g_face = g_new + (P + 1), *g_edge = g_face + (P + 1), *g_pre = g_edge + (P + 1), *g_tab = g_pre + (P + 1);
  HIPCHK(c, hipMemsetAsync(d_ref, 0, (size_t)pre_off[P], c->stream));
  LAUNCH(c, "k_stl_table_init", k_stl_table_init, dim3(256), dim3(256), (int2*)d_tab, (size_t)tab_off[P]);
  LAUNCH(c, "k_clip_emit", k_clip_emit, dim3(gf, P), dim3(256), (const double*)d_v, (const int*)d_f, nf, nv, (const double*)d_pl, (const signed char*)d_sign,
         (const unsigned char*)d_cls, (const int*)d_fpos, (const ClipCounts*)d_cnt, g_new, g_face, g_edge, (double*)d_np, (int*)d_pf, (int*)d_pe);
  LAUNCH(c, "k_clip_mark", k_clip_mark, dim3(gf, P), dim3(256), (const int*)d_pf, g_face, g_pre, (unsigned char*)d_ref);
  LAUNCH(c, "k_clip_hash", k_clip_hash, dim3(gv + gf, P), dim3(256), (const double*)d_v, nv, (const double*)d_np, g_new, g_pre, (const unsigned char*)d_ref,
         (long long*)d_keys, (int2*)d_tab, g_tab, (int*)d_slot);
  LAUNCH(c, "k_clip_rank", k_clip_rank, dim3(P), dim3(SH_STL_SCAN_THREADS), g_pre, (const unsigned char*)d_ref, (const int2*)d_tab, g_tab, (const int*)d_slot,
         (int*)d_vid, (ClipCounts*)d_cnt);
  HIPCHK(c, hipMemcpyAsync(cn.data(), d_cnt, (size_t)P * sizeof(ClipCounts), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  report(false);
  for (int p = 0; p < P; ++p)
    if (cn[p].n_verts > cap_v) return fail(c, SH_ERR_CAPACITY, "sh_slice_mesh_planes: vertex capacity too small (counts hold the sizes needed)");
  LAUNCH(c, "k_clip_out", k_clip_out, dim3(gv + gf, P), dim3(256), (const double*)d_v, nv, (const double*)d_np, g_new, g_pre, (const unsigned char*)d_ref,
         (const int2*)d_tab, g_tab, (const int*)d_slot, (const int*)d_vid, (const int*)d_pf, g_face, (const int*)d_pe, g_edge, (double*)d_ov, cap_v, (int*)d_of,
         cap_f, (int*)d_oe, cap_e);
  for (int p = 0; p < P; ++p) {
    if (cn[p].n_verts) HIPCHK(c, hipMemcpyAsync(out_verts + 3 * (size_t)p * cap_v, (double*)d_ov + 3 * (size_t)p * cap_v, (size_t)cn[p].n_verts * 24, hipMemcpyDeviceToHost, c->stream));
    if (cn[p].n_faces) HIPCHK(c, hipMemcpyAsync(out_faces + 3 * (size_t)p * cap_f, (int*)d_of + 3 * (size_t)p * cap_f, (size_t)cn[p].n_faces * 12, hipMemcpyDeviceToHost, c->stream));
    if (out_edges && cn[p].n_edges) HIPCHK(c, hipMemcpyAsync(out_edges + 2 * (size_t)p * cap_e, (int*)d_oe + 2 * (size_t)p * cap_e, (size_t)cn[p].n_edges * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

int sh_mesh_transformed(sh_ctx* c, int b, const double* T, double* out) {
  if (!c || !T || !out || b < 0 || b >= c->B) return fail(c, SH_ERR_ARG, "sh_mesh_transformed: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  long long V = c->h_voff[b + 1] - c->h_voff[b];
  int rc;
  if ((rc = ensure(c, "mt_T", 16 * 8, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "mt_out", V * 3 * 8, 8)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<double>(c, "mt_T"), T, 16 * 8, hipMemcpyHostToDevice, c->stream));
  dim3 grid((unsigned)std::min<long long>((V + 255) / 256, 1024));
  LAUNCH(c, "k_affine_f32in", k_affine_f32in, grid, dim3(256), buf<double>(c, "mt_T"),
         buf<float>(c, "verts") + 3 * c->h_voff[b], buf<double>(c, "mt_out"), V);
  HIPCHK(c, hipMemcpyAsync(out, buf<double>(c, "mt_out"), V * 3 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

// ---- UNet forward (f32 MFMA path) ----------------------------------------------------------------------
static int conv_layer(sh_ctx* c, const char* lname, const sh_ctx::ULayer& L, const float* src0, const float* src1, int C0, int C1, float* dst,
                      int H, int W, int nimg, int relu, int fuse = 0, float* pooled = nullptr, const float* head_w = nullptr, const float* head_b = nullptr,
                      float* logits = nullptr, const float* image = nullptr, const float* w0 = nullptr, const float* b0 = nullptr) {
  if (H % UN_TH || W % UN_TW) return fail(c, SH_ERR_ARG, "unet: feature map is not a multiple of 16");
  const float* P = buf<float>(c, "params");
  const float* w = P + L.w_off; const float* b = P + L.b_off;
  const int tiles = (H / UN_TH) * (W / UN_TW);
  if (c->params.unet_dtype == SH_UNET_F32X && C0 % 32 == 0 && C1 % 32 == 0 && L.cout % 32 == 0) {
    // split-f16 operands on the 16-bit matrix pipe (k_unet_x3.h); weights split once per parameter block by unet_forward
    const u16* wh = buf<u16>(c, "params_x3h") + L.w_off;
    const u16* wl = buf<u16>(c, "params_x3l") + L.w_off;
    float* np_ = nullptr; const float* nf_ = nullptr;
    if (L.taps == 9 && fuse == (UF_FIRST | UF_POOL) && L.cout == 32 && C0 == 32 && C1 == 0) {      // enc0b with enc0a computed while its halo tile is staged
      LAUNCH(c, lname, (k_conv_mfma_x3<9, 2, UF_FIRST | UF_POOL, 0>), dim3(tiles, 1, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, pooled, nf_, nf_, np_, image, w0, b0);
    }
    else if (L.taps == 9 && fuse == UF_HEAD && L.cout == 32) { LAUNCH(c, lname, (k_conv_mfma_x3<9, 2, UF_HEAD>), dim3(tiles, 1, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, np_, head_w, head_b, logits, nf_, nf_, nf_); }
    else if (L.taps == 9 && fuse == UF_POOL && L.cout % 64 == 0) { LAUNCH(c, lname, (k_conv_mfma_x3<9, 4, UF_POOL>), dim3(tiles, L.cout / 64, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, pooled, nf_, nf_, np_, nf_, nf_, nf_); }
    else if (L.taps == 9 && fuse == UF_POOL) { LAUNCH(c, lname, (k_conv_mfma_x3<9, 2, UF_POOL, 0>), dim3(tiles, L.cout / 32, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, pooled, nf_, nf_, np_, nf_, nf_, nf_); }
    else if (fuse != 0) return fail(c, SH_ERR_ARG, "unet: unsupported fusion");
    else if (L.taps == 9 && L.cout % 64 == 0) { LAUNCH(c, lname, (k_conv_mfma_x3<9, 4>), dim3(tiles, L.cout / 64, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, np_, nf_, nf_, np_, nf_, nf_, nf_); }
    else if (L.taps == 9) { LAUNCH(c, lname, (k_conv_mfma_x3<9, 2, 0, 0>), dim3(tiles, L.cout / 32, nimg), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, relu, np_, nf_, nf_, np_, nf_, nf_, nf_); }
    else if (C1 == 0 && W % 32 == 0 && H % 16 == 0 && (C0 == 64 || C0 == 128 || C0 == 256 || C0 == 512)) {
      // up-convolutions with the source pixels resident in registers (k_upconv_x3r)
      if (C0 == 64) { LAUNCH(c, lname, (k_upconv_x3r<2, 4>), dim3((W / 32) * (H / 16), nimg), dim3(UXR_THREADS), src0, wh, wl, b, dst, H, W, L.cout); }
      else if (C0 == 128) { LAUNCH(c, lname, (k_upconv_x3r<4, 4>), dim3((W / 32) * (H / 16), nimg), dim3(UXR_THREADS), src0, wh, wl, b, dst, H, W, L.cout); }
      else if (C0 == 256) { LAUNCH(c, lname, (k_upconv_x3r<8, 2>), dim3((W / 32) * (H / 8), nimg), dim3(UXR_THREADS), src0, wh, wl, b, dst, H, W, L.cout); }
      else { LAUNCH(c, lname, (k_upconv_x3r<16, 1>), dim3((W / 32) * (H / 4), nimg), dim3(UXR_THREADS), src0, wh, wl, b, dst, H, W, L.cout); }
    }
    else if (L.cout % 64 == 0) { LAUNCH(c, lname, (k_conv_mfma_x3<1, 4>), dim3(tiles, L.cout / 64, nimg * 4), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, 0, np_, nf_, nf_, np_, nf_, nf_, nf_); }
    else { LAUNCH(c, lname, (k_conv_mfma_x3<1, 2>), dim3(tiles, L.cout / 32, nimg * 4), dim3(UN_THREADS), src0, src1, C0, C1, wh, wl, b, dst, H, W, L.cout, 0, np_, nf_, nf_, np_, nf_, nf_, nf_); }
    return SH_OK;
  }
  if (L.taps == 9) {
    if (L.cout % 64 == 0) {
      LAUNCH(c, lname, (k_conv_mfma_f32<9, 4>), dim3(tiles, L.cout / 64, nimg), dim3(UN_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu);
    } else {
      LAUNCH(c, lname, (k_conv_mfma_f32<9, 2>), dim3(tiles, L.cout / 32, nimg), dim3(UN_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu);
    }
  } else {
    if (L.cout % 64 == 0) {
      LAUNCH(c, lname, (k_conv_mfma_f32<1, 4>), dim3(tiles, L.cout / 64, nimg * 4), dim3(UN_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, 0);
    } else {
      LAUNCH(c, lname, (k_conv_mfma_f32<1, 2>), dim3(tiles, L.cout / 32, nimg * 4), dim3(UN_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, 0);
    }
  }
  return SH_OK;
}

// ---- UNet turns ----------------------------------------------------------------------------------------
// Several contexts on one device overlap well when the launch-bound geometry kernels of one run beside the chip-filling
// UNet kernels of another -- and badly when two UNet passes share the CUs (each just takes twice as long).  Contexts
// that opted in (sh_set_unet_turns) therefore chain their UNet passes with events, in the order the host enqueued them.
static std::mutex g_turn_mu;
static hipEvent_t g_turn_last[64] = {};      // per device: recorded at the end of the most recently enqueued UNet pass
static sh_ctx* g_turn_owner[64] = {};

static int unet_turn_enter(sh_ctx* c) {
  if (!c->unet_turn || c->device < 0 || c->device >= 64) return SH_OK;
  std::lock_guard<std::mutex> lk(g_turn_mu);
  if (g_turn_last[c->device] && g_turn_owner[c->device] != c) HIPCHK(c, hipStreamWaitEvent(c->stream, g_turn_last[c->device], 0));
  return SH_OK;
}

static int unet_turn_leave(sh_ctx* c) {
  if (!c->unet_turn || c->device < 0 || c->device >= 64) return SH_OK;
  std::lock_guard<std::mutex> lk(g_turn_mu);
  if (!c->unet_done_ev) HIPCHK(c, hipEventCreateWithFlags(&c->unet_done_ev, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->unet_done_ev, c->stream));
  g_turn_last[c->device] = c->unet_done_ev;
  g_turn_owner[c->device] = c;
  return SH_OK;
}

static void unet_turn_forget(sh_ctx* c) {
  std::lock_guard<std::mutex> lk(g_turn_mu);
  if (c->device >= 0 && c->device < 64 && g_turn_owner[c->device] == c) { g_turn_last[c->device] = nullptr; g_turn_owner[c->device] = nullptr; }
}

static int unet_forward(sh_ctx* c, const float* image, float* logits, int nimg, int H, int W) {
  const int D = c->unet_depth, base = c->unet_base;
  if ((H >> D) % 16 || (W >> D) % 16) return fail(c, SH_ERR_ARG, "unet: input size must be a multiple of 16 << depth");
  int rc;
  const size_t full = (size_t)nimg * H * W * base * 4;
  if ((rc = ensure(c, "unet.a", full, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "unet.b", full, 4)) != SH_OK) return rc;
  std::vector<float*> skip(D);
  for (int i = 0; i < D; ++i) {
    std::string nm = "unet.skip" + std::to_string(i);
    if ((rc = ensure(c, nm.c_str(), full >> i, 4)) != SH_OK) return rc;     // H*W/4^i * base*2^i
    skip[i] = buf<float>(c, nm.c_str());
  }
  float* A = buf<float>(c, "unet.a");
  float* Bq = buf<float>(c, "unet.b");
  const float* P = buf<float>(c, "params");
  if (c->params.unet_dtype == SH_UNET_F32X) {      // split the MFMA layers' weights into f16 high / low parts: one launch, once per parameter block
    if ((rc = ensure(c, "params_x3h", c->unet_floats * 2, 2)) != SH_OK) return rc;
    if ((rc = ensure(c, "params_x3l", c->unet_floats * 2, 2)) != SH_OK) return rc;
    if (!c->packed_x3) {
      std::vector<PackEntry> tab;
      long long total = 0;
      for (auto& kv : c->ulayers) {
        const sh_ctx::ULayer& l = kv.second;
        if (l.cin < 32 || l.cout < 32) continue;
        tab.push_back(PackEntry{total, (long long)l.w_off, l.taps, l.cin, l.cout, 0});
        total += (long long)l.taps * l.cin * l.cout;
        // range of the split: 64 w must be a finite f16 (|w| < 65504 / 64); beyond it the high part is an infinity and the layer's
        // outputs NaN, silently (include/shoulder_hip.h, SH_UNET_F32X)
        if (c->h_unet.size() >= l.w_off + (size_t)l.taps * l.cin * l.cout) {
          const float* wl = c->h_unet.data() + l.w_off;
          for (size_t i = 0, n = (size_t)l.taps * l.cin * l.cout; i < n; ++i)
            if (!(fabsf(wl[i]) < 65504.0f / X3_WSCALE)) {
              char m[200];
              snprintf(m, sizeof m, "SH_UNET_F32X: layer %s has a weight of magnitude %g; the split-f16 operands hold |w| < %g (use SH_UNET_F32 for this network)",
                       kv.first.c_str(), (double)fabsf(wl[i]), (double)(65504.0f / X3_WSCALE));
              return fail(c, SH_ERR_ARG, m);
            }
        }
      }
      if ((rc = ensure(c, "unet16.packtab", tab.size() * sizeof(PackEntry), 8)) != SH_OK) return rc;
      HIPCHK(c, hipMemcpyAsync(c->bufs["unet16.packtab"].p, tab.data(), tab.size() * sizeof(PackEntry), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));      // `tab` is a local
      c->packtab_ready = true;
      LAUNCH(c, "k_pack_w_x3", k_pack_w_x3, dim3(2048), dim3(256), P, buf<u16>(c, "params_x3h"), buf<u16>(c, "params_x3l"), (const PackEntry*)c->bufs["unet16.packtab"].p, (int)tab.size(), total);
      c->packed_x3 = true;
    }
  }
  auto L = [&](const std::string& n) -> const sh_ctx::ULayer& { return c->ulayers[n]; };
  int h = H, w = W;
  // SH_UNET_F32X: the 2x2 pools ride in the epilogue of the conv before them (k_unet_x3.h), and with 32 base channels the first
  // conv is computed inside enc0b's staging
  const bool x3 = c->params.unet_dtype == SH_UNET_F32X && base % 32 == 0;
  const bool x3_first = x3 && base == 32;
  if (!x3_first) {
    const sh_ctx::ULayer& l = L("enc0a");
    size_t npx = (size_t)nimg * h * w;
    LAUNCH(c, "unet.enc0a", k_conv_first, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 8192)), dim3(256), image, P + l.w_off, P + l.b_off, A, h, w, l.cout, nimg);
  }
  if (x3_first) {
    const sh_ctx::ULayer& l = L("enc0a");
    if ((rc = conv_layer(c, "unet.enc0b", L("enc0b"), A, nullptr, base, 0, skip[0], h, w, nimg, 1, UF_FIRST | UF_POOL, Bq, nullptr, nullptr, nullptr, image, P + l.w_off, P + l.b_off)) != SH_OK) return rc;
  } else if ((rc = conv_layer(c, "unet.enc0b", L("enc0b"), A, nullptr, base, 0, skip[0], h, w, nimg, 1, x3 ? UF_POOL : 0, Bq)) != SH_OK) return rc;
  if (x3) std::swap(A, Bq);      // (the pooled tensor is the next level's input, which the loop below reads from A)
  int ch = base;
  for (int i = 1; i <= D; ++i) {
    if (!x3) {
      size_t e = (size_t)nimg * (h / 2) * (w / 2) * (ch / 4);
      LAUNCH(c, "unet.pool", k_maxpool2, dim3((unsigned)std::min<size_t>((e + 255) / 256, 8192)), dim3(256), skip[i - 1], A, h, w, ch, nimg);
    }
    h /= 2; w /= 2;
    std::string na = i < D ? "enc" + std::to_string(i) + "a" : "bota", nb = i < D ? "enc" + std::to_string(i) + "b" : "botb";
    if ((rc = conv_layer(c, ("unet." + na).c_str(), L(na), A, nullptr, ch, 0, Bq, h, w, nimg, 1)) != SH_OK) return rc;
    ch *= 2;
    float* dst = i < D ? skip[i] : A;
    // (A was consumed by the conv above: with the fused pool it receives the next level's input)
    if ((rc = conv_layer(c, ("unet." + nb).c_str(), L(nb), Bq, nullptr, ch, 0, dst, h, w, nimg, 1, (x3 && i < D) ? UF_POOL : 0, A)) != SH_OK) return rc;
  }
  // decoder: x lives in A
  float* x = A; float* y = Bq;
  for (int i = D - 1; i >= 0; --i) {
    std::string nu = "up" + std::to_string(i), na = "dec" + std::to_string(i) + "a", nb = "dec" + std::to_string(i) + "b";
    if ((rc = conv_layer(c, ("unet." + nu).c_str(), L(nu), x, nullptr, ch, 0, y, h, w, nimg, 0)) != SH_OK) return rc;
    h *= 2; w *= 2; ch /= 2;
    if ((rc = conv_layer(c, ("unet." + na).c_str(), L(na), skip[i], y, ch, ch, x, h, w, nimg, 1)) != SH_OK) return rc;
    // (the head stays on k_head: its sequential f32 chain over the channels is the exact path's; fused into dec0b's epilogue the
    //  logits move by another ~1e-6 and one mask pixel of the 64-humerus bench batch flips)
    if ((rc = conv_layer(c, ("unet." + nb).c_str(), L(nb), x, nullptr, ch, 0, y, h, w, nimg, 1)) != SH_OK) return rc;
    std::swap(x, y);
  }
  {
    const sh_ctx::ULayer& l = L("head");
    size_t npx = (size_t)nimg * H * W;
    if (l.cin <= 32) { LAUNCH(c, "unet.head", k_head<32>, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 16384)), dim3(256), x, P + l.w_off, P + l.b_off, logits, l.cin, npx); }
    else { LAUNCH(c, "unet.head", k_head<64>, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 8192)), dim3(256), x, P + l.w_off, P + l.b_off, logits, l.cin, npx); }
  }
  return SH_OK;
}

}  // extern "C" (the templates below need C++ linkage)

// ---- UNet forward (16-bit MFMA paths: EK = 0 __bf16, 1 _Float16; tensors as raw u16) -----------------------------------------
#define SH_UNET_TICKETS 64
#define SH_UNET_TKTAB (1 << 18)
// Workgroups of a persistent UNet launch.  Each takes a whole CU (its LDS, all of its registers), so while one is resident no
// other kernel can start there: beside the UNet pass of one lane, every launch of the other lane's geometry chain (~40 per step)
// waited ~50 us for a workgroup to end, and the chain took 7-8 ms instead of 3.2.  Contexts that take turns on a device
// (sh_set_unet_turns: there IS another lane) therefore leave SHOULDER_CU_RESERVE CUs (default 32 = 4 per XCD; read once) out of
// the grid; the work tickets spread the items over whatever grid there is.  Measured on the two-lane headline: 0 / 8 / 16 / 32 /
// 48 / 64 / 96 reserved -> 8.72 / 8.80 / 8.73 / 8.27 / 8.54 / 8.56 / 9.35 ms per step.
static int cu_reserve() {
  static const int reserve = getenv("SHOULDER_CU_RESERVE") ? std::max(0, atoi(getenv("SHOULDER_CU_RESERVE"))) : 32;
  return reserve;
}
static int persistent_grid(sh_ctx* c) {
  if (c->num_cus <= 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) v = 0; c->num_cus = v > 0 ? v : 256; }
  return c->unet_turn ? std::max(8, c->num_cus - cu_reserve()) : c->num_cus;
}

// work tickets of a persistent launch: the next free counter of this forward pass and the table of item bounds of runs of decreasing
// length for (items, workgroups, cout groups) -- every ticket a third of what would be a fair share of the remaining items, whole
// cout-group sets of a tile (its input tile comes from HBM once) -- built once per shape
static int unet_tickets(sh_ctx* c, int total, int nwg, int ngrp, unsigned** tk, const int** tk_tab, int* ntk) {
  const auto key = std::make_tuple(total, nwg, ngrp);
  auto it = c->tk_tabs.find(key);
  if (it == c->tk_tabs.end()) {
    std::vector<int> tab;
    int pos = 0;
    while (pos < total) {
      int sz = std::max(1, (int)std::ceil((total - pos) / (3.0 * (double)nwg)));
      if (sz >= ngrp) sz = sz / ngrp * ngrp;
      tab.push_back(pos);
      pos += std::min(sz, total - pos);
    }
    tab.push_back(total);
    if ((int)tab.size() > SH_UNET_TKTAB) return fail(c, SH_ERR_CAPACITY, "unet: ticket table larger than its buffer");
    if (c->tk_tab_used + (int)tab.size() > SH_UNET_TKTAB) {      // many different shapes (sh_unet_infer with varying n): start the cache over
      HIPCHK(c, hipStreamSynchronize(c->stream));                  // (launches that read the old tables are done)
      c->tk_tabs.clear();
      c->tk_tab_used = 0;
    }
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "unet16.tk_tab") + c->tk_tab_used, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // `tab` is a local
    it = c->tk_tabs.emplace(key, std::make_pair(c->tk_tab_used, (int)tab.size() - 1)).first;
    c->tk_tab_used += (int)tab.size();
  }
  if (c->ticket_next >= SH_UNET_TICKETS) return fail(c, SH_ERR_CAPACITY, "unet: out of work counters");
  *tk = buf<unsigned>(c, "unet16.tickets") + c->ticket_next++;
  *tk_tab = buf<int>(c, "unet16.tk_tab") + it->second.first;
  *ntk = it->second.second;
  return SH_OK;
}

// One layer of the 16-bit network.  3x3 convs with a multiple of 64 output channels on 32 x 16-tileable maps run on the persistent
// LDS-DMA kernel (k_unet16_ldr.h; UF_POOL: the 2x2 max pool written beside the output); 2x2 transposed convs on k_upconv16g /
// k_upconv16; everything else on the generic two-barrier kernel k_conv_mfma16 (k_unet_bf16.h), which is also the whole of the
// REFERENCE network (sh_ctx::unet_reference: layer by layer, nothing fused, no persistent kernel -- what the tests hold the
// production kernels against).
template <int EK>
static int conv_layer16(sh_ctx* c, const char* lname, const sh_ctx::ULayer& L, const u16* src0, const u16* src1, int C0, int C1,
                           u16* dst, int H, int W, int nimg, int relu, int fuse = 0, ConvFuse fz = ConvFuse{}, const UpSrc* upsrc = nullptr /*the second
                           source is up(low), computed by the conv's loader waves (k_unet16_ldr.h): C1 = its channels, 128 low-resolution channels*/) {
  if (H % UN_TH || W % UN_TW) return fail(c, SH_ERR_ARG, "unet: feature map is not a multiple of 16");
  const u16* w = buf<u16>(c, "params_bf16") + L.w_off;
  const float* b = buf<float>(c, "params") + L.b_off;
  const int tiles = (H / UN_TH) * (W / UN_TW);
  const dim3 blk(UN_THREADS);
  const bool ldr = !c->unet_reference && L.taps == 9 && L.cout % 64 == 0 && L.cout <= 512 && W % 32 == 0 && H % 16 == 0 && C0 % 32 == 0 && C1 % 32 == 0 &&
                   (fuse == 0 || (fuse == UF_POOL && relu));      // (its fused pool works on ReLU'd values)
  if (ldr) {
    int rc0;
    if ((rc0 = ensure(c, "unet16.zero", 256, 2)) != SH_OK) return rc0;
    if (!c->zero_page_ready) { HIPCHK(c, hipMemsetAsync(buf<char>(c, "unet16.zero"), 0, 256, c->stream)); c->zero_page_ready = true; }
    const int total = nimg * (W / 32) * (H / 16) * (L.cout / 64);
    const dim3 g((unsigned)std::min(total, persistent_grid(c)));
    unsigned* tk = nullptr; const int* tk_tab = nullptr; int ntk = 0;
    if ((rc0 = unet_tickets(c, total, (int)g.x, L.cout / 64, &tk, &tk_tab, &ntk)) != SH_OK) return rc0;
    const u16* zp = (const u16*)c->bufs["unet16.zero"].p;
    u16* pl = fuse == UF_POOL ? (u16*)fz.pooled : (u16*)nullptr;
    // weights resident in LDS: one cout group whose packed weights fit behind the two input buffers (32 -> 64 and 64 -> 64 layers)
    const bool wres = L.cout == 64 && ((C0 + C1) / 32) * 64 <= 128;
    const UpSrc nou{nullptr, nullptr, nullptr};
    if (upsrc) {
      if (fuse != 0 || wres || L.cout != 64) return fail(c, SH_ERR_ARG, "unet: an up-convolution inside this conv shape is not built");
      LAUNCH(c, lname, (k_conv3_ldr16<EK, 0, 0, 4>), g, dim3(UD_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, nimg, zp, pl, tk, tk_tab, ntk, *upsrc);
    }
    else if (fuse == UF_POOL && wres) { LAUNCH(c, lname, (k_conv3_ldr16<EK, UF_POOL, 1>), g, dim3(UD_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, nimg, zp, pl, tk, tk_tab, ntk, nou); }
    else if (fuse == UF_POOL) { LAUNCH(c, lname, (k_conv3_ldr16<EK, UF_POOL, 0>), g, dim3(UD_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, nimg, zp, pl, tk, tk_tab, ntk, nou); }
    else if (wres) { LAUNCH(c, lname, (k_conv3_ldr16<EK, 0, 1>), g, dim3(UD_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, nimg, zp, pl, tk, tk_tab, ntk, nou); }
    else { LAUNCH(c, lname, (k_conv3_ldr16<EK, 0, 0>), g, dim3(UD_THREADS), src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, nimg, zp, pl, tk, tk_tab, ntk, nou); }
  } else if (L.taps == 9 && L.cout % 64 == 0) {
    const dim3 g(tiles, L.cout / 64, nimg);
    if (fuse == 0) { LAUNCH(c, lname, (k_conv_mfma16<EK, 9, 4, 0>), g, blk, src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, fz); }
    else if (fuse == UF_POOL) { LAUNCH(c, lname, (k_conv_mfma16<EK, 9, 4, UF_POOL>), g, blk, src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, fz); }
    else return fail(c, SH_ERR_ARG, "unet: unsupported fusion");
  } else if (L.taps == 9) {
    const dim3 g(tiles, L.cout / 32, nimg);
    if (fuse == 0) { LAUNCH(c, lname, (k_conv_mfma16<EK, 9, 2, 0>), g, blk, src0, src1, C0, C1, w, b, dst, H, W, L.cout, relu, fz); }
    else return fail(c, SH_ERR_ARG, "unet: unsupported fusion");
  } else if (!c->unet_reference && L.cout % 32 == 0 && C1 == 0 && C0 % 32 == 0) {
    // 2x2 transposed conv (k_unet16_up.h): source pixels in registers, the weights of a 32-cout group by LDS-DMA, one barrier per
    // group (Cin = 512: per two phases); the staged form otherwise
    const bool upg = W % 32 == 0 && H % 16 == 0 && (C0 == 128 || C0 == 256 || C0 == 512) && L.cout <= 512;
    if (upg) {
      // items = (image, source tile of 32 x 4 MT pixels) on the grid of the persistent convolutions, handed out by work tickets; up3 has
      // about one item per CU: one workgroup per item
      const int mt = C0 == 128 ? 4 : 2, nitems = (W / 32) * (H / (4 * mt)) * nimg;
      const int grid = C0 == 512 ? nitems : std::min(nitems, persistent_grid(c));
      unsigned* tk = nullptr; const int* tk_tab = nullptr; int ntk = 0;
      if (C0 != 512) { const int trc = unet_tickets(c, nitems, grid, 1, &tk, &tk_tab, &ntk); if (trc != SH_OK) return trc; }
      if (C0 == 128) { LAUNCH(c, lname, (k_upconv16g<EK, 4, 4, 4, true>), dim3((unsigned)grid), dim3(UPR_THREADS), src0, w, b, dst, H, W, L.cout, nimg, tk, tk_tab, ntk); }
      else if (C0 == 256) { LAUNCH(c, lname, (k_upconv16g<EK, 8, 2, 4, true>), dim3((unsigned)grid), dim3(UPR_THREADS), src0, w, b, dst, H, W, L.cout, nimg, tk, tk_tab, ntk); }
      else { LAUNCH(c, lname, (k_upconv16g<EK, 16, 2, 2, false>), dim3((unsigned)grid), dim3(UPR_THREADS), src0, w, b, dst, H, W, L.cout, nimg, tk, tk_tab, ntk); }
    }
    else { LAUNCH(c, lname, (k_upconv16<EK>), dim3(tiles, L.cout / 32, nimg * 2), dim3(UPC_THREADS), src0, C0, w, b, dst, H, W, L.cout); }
  } else if (L.cout % 64 == 0) {
    LAUNCH(c, lname, (k_conv_mfma16<EK, 1, 4, 0>), dim3(tiles, L.cout / 64, nimg * 4), blk, src0, src1, C0, C1, w, b, dst, H, W, L.cout, 0, fz);
  } else {
    LAUNCH(c, lname, (k_conv_mfma16<EK, 1, 2, 0>), dim3(tiles, L.cout / 32, nimg * 4), blk, src0, src1, C0, C1, w, b, dst, H, W, L.cout, 0, fz);
  }
  return SH_OK;
}

// does the 16-bit forward run its fused level-0 kernels (k_unet16_pp.h)?  (run_window asks: k_enc0_pp can read the unscaled image)
static bool unet16_level0_fused(const sh_ctx* c, int H, int W) {
  return !c->unet_reference && c->unet_base == 32 && c->unet_depth >= 1 && W % 32 == 0 && H % 16 == 0 && (H >> c->unet_depth) % 16 == 0 && (W >> c->unet_depth) % 16 == 0;
}
static bool unet16_starts_fused(const sh_ctx* c, int H, int W) { return unet16_level0_fused(c, H, W); }

// Double-conv UNet, 16-bit.  With 32 base channels the full-resolution level runs as three fused ping-pong kernels (k_unet16_pp.h:
// image -> enc0a -> enc0b -> skip0 + pool; up0 + dec0a; dec0b + head) and every 2x2 max pool rides in the epilogue of the conv before
// it.  Other widths, maps that do not tile, and the reference network run layer by layer.
template <int EK>
static int unet_forward16(sh_ctx* c, const float* image, float* logits, int nimg, int H, int W) {
  const int D = c->unet_depth, base = c->unet_base;
  if ((H >> D) % 16 || (W >> D) % 16) return fail(c, SH_ERR_ARG, "unet: input size must be a multiple of 16 << depth");
  int rc;
  if ((rc = ensure(c, "params_bf16", c->unet_floats * 2, 2)) != SH_OK) return rc;
  const float* P = buf<float>(c, "params");
  u16* PW = buf<u16>(c, "params_bf16");
  if (c->packed_kind != EK) {     // pack the MFMA layers' weights for this element type: one launch for all layers, once per parameter block
    std::vector<PackEntry> tab;
    long long total = 0;
    for (auto& kv : c->ulayers) {
      const sh_ctx::ULayer& l = kv.second;
      if (l.cin < 32 || l.cout < 32) continue;
      tab.push_back(PackEntry{total, (long long)l.w_off, l.taps, l.cin, l.cout, 0});
      total += (long long)l.taps * l.cin * l.cout;
    }
    if ((rc = ensure(c, "unet16.packtab", tab.size() * sizeof(PackEntry), 8)) != SH_OK) return rc;
    if (!c->packtab_ready) {
      HIPCHK(c, hipMemcpyAsync(c->bufs["unet16.packtab"].p, tab.data(), tab.size() * sizeof(PackEntry), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));      // `tab` is a local
      c->packtab_ready = true;
    }
    LAUNCH(c, "k_pack_w_bf16", k_pack_w16_all<EK>, dim3(2048), dim3(256), P, PW, (const PackEntry*)c->bufs["unet16.packtab"].p, (int)tab.size(), total);
    c->packed_kind = EK;
  }
  if ((rc = ensure(c, "unet16.tickets", SH_UNET_TICKETS * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "unet16.tk_tab", SH_UNET_TKTAB * 4, 4)) != SH_OK) return rc;
  FILL(c, {buf<unsigned>(c, "unet16.tickets"), (size_t)SH_UNET_TICKETS * 4, 0});
  c->ticket_next = 0;
  const bool fused = unet16_level0_fused(c, H, W);      // level 0 on the ping-pong kernels, pools in the conv epilogues
  const size_t full = (size_t)nimg * H * W * base * 2;
  if ((rc = ensure(c, "unet16.a", full, 2)) != SH_OK) return rc;
  if ((rc = ensure(c, "unet16.b", full, 2)) != SH_OK) return rc;
  std::vector<u16*> skip(D);
  for (int i = 0; i < D; ++i) {
    std::string nm = "unet16.skip" + std::to_string(i);
    if ((rc = ensure(c, nm.c_str(), full >> i, 2)) != SH_OK) return rc;
    skip[i] = buf<u16>(c, nm.c_str());
  }
  u16* A = buf<u16>(c, "unet16.a");
  u16* Bq = buf<u16>(c, "unet16.b");
  auto L = [&](const std::string& n) -> const sh_ctx::ULayer& { return c->ulayers[n]; };
  if ((rc = ensure(c, "unet16.zero", 256, 2)) != SH_OK) return rc;
  if (!c->zero_page_ready) { HIPCHK(c, hipMemsetAsync(buf<char>(c, "unet16.zero"), 0, 256, c->stream)); c->zero_page_ready = true; }
  const u16* zp = (const u16*)c->bufs["unet16.zero"].p;
  int h = H, w = W;
  if (fused) {
    // level-0 encoder (k_enc0_pp): image -> enc0a -> LDS -> enc0b -> skip0 + pooled
    const sh_ctx::ULayer& la = L("enc0a");
    const sh_ctx::ULayer& lb = L("enc0b");
    const int total = nimg * (w / 32) * (h / 16);
    const unsigned grid = (unsigned)std::min(total, persistent_grid(c));
    unsigned* tk = nullptr; const int* tk_tab = nullptr; int ntk = 0;
    if ((rc = unet_tickets(c, total, (int)grid, 1, &tk, &tk_tab, &ntk)) != SH_OK) return rc;
    LAUNCH_FN(c, "unet.enc0b", launch_enc0_pp(EK, grid, c->stream, image, P + la.w_off, P + la.b_off, PW + lb.w_off, P + lb.b_off, skip[0], A, h, w, nimg,
                                              c->unet_raw, c->unet_mm, tk, tk_tab, ntk));
  } else {
    const sh_ctx::ULayer& l = L("enc0a");
    size_t npx = (size_t)nimg * h * w;
    LAUNCH(c, "unet.enc0a", k_conv_first16<EK>, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 8192)), dim3(256), image, P + l.w_off, P + l.b_off, A, h, w, l.cout, nimg);
    if ((rc = conv_layer16<EK>(c, "unet.enc0b", L("enc0b"), A, nullptr, base, 0, skip[0], h, w, nimg, 1)) != SH_OK) return rc;
  }
  int ch = base;
  for (int i = 1; i <= D; ++i) {
    if (!fused) {
      size_t e = (size_t)nimg * (h / 2) * (w / 2) * (ch / 8);
      LAUNCH(c, "unet.pool", k_maxpool2_16<EK>, dim3((unsigned)std::min<size_t>((e + 255) / 256, 8192)), dim3(256), skip[i - 1], A, h, w, ch, nimg);
    }
    h /= 2; w /= 2;
    std::string na = i < D ? "enc" + std::to_string(i) + "a" : "bota", nb = i < D ? "enc" + std::to_string(i) + "b" : "botb";
    if ((rc = conv_layer16<EK>(c, ("unet." + na).c_str(), L(na), A, nullptr, ch, 0, Bq, h, w, nimg, 1)) != SH_OK) return rc;
    ch *= 2;
    u16* dst = i < D ? skip[i] : A;
    ConvFuse fz{};
    fz.pooled = A;      // (A was consumed by the conv above; the next level reads it)
    if ((rc = conv_layer16<EK>(c, ("unet." + nb).c_str(), L(nb), Bq, nullptr, ch, 0, dst, h, w, nimg, 1, (fused && i < D) ? UF_POOL : 0, fz)) != SH_OK) return rc;
  }
  u16* x = A; u16* y = Bq;
  for (int i = D - 1; i >= 0; --i) {
    std::string nu = "up" + std::to_string(i), na = "dec" + std::to_string(i) + "a", nb = "dec" + std::to_string(i) + "b";
    if (fused && i == 0) {
      // level 0: the up-convolution computed inside dec0a (k_dec0a_up_pp: x = low-resolution input, y = dec0a's output), then
      // dec0b with the 1x1 head in its epilogue (k_dec0b_head_pp: only the logits leave the kernel)
      h *= 2; w *= 2; ch /= 2;
      const sh_ctx::ULayer& lu = L(nu);
      const sh_ctx::ULayer& la = L(na);
      const sh_ctx::ULayer& lb = L(nb);
      const sh_ctx::ULayer& lh = L("head");
      {
        const int total = nimg * (w / 32) * (h / 8);
        const unsigned grid = (unsigned)std::min(total, persistent_grid(c));
        unsigned* tk = nullptr; const int* tk_tab = nullptr; int ntk = 0;
        if ((rc = unet_tickets(c, total, (int)grid, 1, &tk, &tk_tab, &ntk)) != SH_OK) return rc;
        LAUNCH_FN(c, "unet.dec0a", launch_dec0a_up_pp(EK, grid, c->stream, skip[0], x, PW + la.w_off, P + la.b_off, PW + lu.w_off, P + lu.b_off, y, h, w, nimg,
                                                      zp, tk, tk_tab, ntk));
      }
      {
        const int total = nimg * (w / 32) * (h / 16);
        const unsigned grid = (unsigned)std::min(total, persistent_grid(c));
        unsigned* tk = nullptr; const int* tk_tab = nullptr; int ntk = 0;
        if ((rc = unet_tickets(c, total, (int)grid, 1, &tk, &tk_tab, &ntk)) != SH_OK) return rc;
        LAUNCH_FN(c, "unet.dec0b", launch_dec0b_head_pp(EK, grid, c->stream, y, PW + lb.w_off, P + lb.b_off, P + lh.w_off, P + lh.b_off, logits, h, w, nimg,
                                                        zp, tk, tk_tab, ntk));
      }
      return SH_OK;
    }
    // level 1 (128 -> 64 channels up, then 64 + 64 -> 64), SHOULDER_UP_INSIDE=1: the up-convolution is computed by dec1a's loader waves
    // where its halo chunks are needed (k_conv3_ldr16<.., UPL = 4>: ONE cout group, so every chunk is computed once per tile) -- no up1
    // launch, no up1 tensor
    const bool up_inside = fused && ch == 128 && L(na).cout == 64 && L(nu).cout == 64 && (2 * w) % 32 == 0 && (2 * h) % 16 == 0 && c->sw.up_inside;
    if (up_inside) {
      const UpSrc us{x, buf<u16>(c, "params_bf16") + L(nu).w_off, P + L(nu).b_off};
      h *= 2; w *= 2; ch /= 2;
      if ((rc = conv_layer16<EK>(c, ("unet." + na).c_str(), L(na), skip[i], nullptr, ch, ch, y, h, w, nimg, 1, 0, ConvFuse{}, &us)) != SH_OK) return rc;
      if ((rc = conv_layer16<EK>(c, ("unet." + nb).c_str(), L(nb), y, nullptr, ch, 0, x, h, w, nimg, 1)) != SH_OK) return rc;
      continue;      // (the level's result is in x again)
    }
    if ((rc = conv_layer16<EK>(c, ("unet." + nu).c_str(), L(nu), x, nullptr, ch, 0, y, h, w, nimg, 0)) != SH_OK) return rc;
    h *= 2; w *= 2; ch /= 2;
    if ((rc = conv_layer16<EK>(c, ("unet." + na).c_str(), L(na), skip[i], y, ch, ch, x, h, w, nimg, 1)) != SH_OK) return rc;
    if ((rc = conv_layer16<EK>(c, ("unet." + nb).c_str(), L(nb), x, nullptr, ch, 0, y, h, w, nimg, 1)) != SH_OK) return rc;
    std::swap(x, y);
  }
  {
    const sh_ctx::ULayer& l = L("head");
    size_t npx = (size_t)nimg * H * W;
    LAUNCH(c, "unet.head", k_head16<EK>, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 8192)), dim3(256), x, P + l.w_off, P + l.b_off, logits, l.cin, npx, (size_t)H * W);
  }
  return SH_OK;
}

static int unet_dispatch(sh_ctx* c, const float* image, float* logits, int nimg, int H, int W) {
  switch (c->params.unet_dtype) {
    case SH_UNET_BF16: return unet_forward16<0>(c, image, logits, nimg, H, W);
    case SH_UNET_F16: return unet_forward16<1>(c, image, logits, nimg, H, W);
    default: return unet_forward(c, image, logits, nimg, H, W);
  }
}

extern "C" {

// The network alone (the `ort.InferenceSession.run` call of anatomic_neck.py:67-76): n images
// [n][H][W] float32 on the host -> logits [n][H][W] float32 on the host, in the precision sh_params.unet_dtype selects.
int sh_unet_infer(sh_ctx* c, const float* images, int n, int H, int W, float* logits) {
  if (!c || !images || !logits || n <= 0 || H <= 0 || W <= 0) return fail(c, SH_ERR_ARG, "sh_unet_infer: bad argument");
  if (c->ulayers.empty()) return fail(c, SH_ERR_STATE, "sh_unet_infer: no UNet weights loaded");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)n * H * W * 4;
  int rc;
  if ((rc = ensure(c, "infer.image", bytes, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "infer.logits", bytes, 4)) != SH_OK) return rc;
  const int b0 = c->b0; c->b0 = 0;      // named buffers below are whole-batch
  HIPCHK(c, hipMemcpyAsync(buf<float>(c, "infer.image"), images, bytes, hipMemcpyHostToDevice, c->stream));
  rc = unet_dispatch(c, buf<float>(c, "infer.image"), buf<float>(c, "infer.logits"), n, H, W);
  c->b0 = b0;
  if (rc != SH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
  HIPCHK(c, hipMemcpyAsync(logits, buf<float>(c, "infer.logits"), bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

// ---- stage runner ----------------------------------------------------------------------------------
// ---- overflow planes of the slice layer (k_ovf.h) --------------------------------------------------------
static int ovf_pools(sh_ctx* c, OvfPools* P) {
  int rc;
  if ((rc = ensure(c, "ovf.segs", c->ovf_seg_cap * sizeof(Seg), 1)) != SH_OK) return rc;
  if ((rc = ensure(c, "ovf.ring", c->ovf_ring_cap * 16, 8)) != SH_OK) return rc;
  if ((rc = ensure(c, "ovf.work", c->ovf_work_cap, 1)) != SH_OK) return rc;
  if ((rc = ensure(c, "ovf.ctr", 128, 8)) != SH_OK) return rc;      // words 0..7: the slice layer's pools (k_ovf.h) and the end sections; 8, 9: k_obb_candidates (ObbWs::need)
  for (const char* n : {"ovf.segs", "ovf.ring", "ovf.work", "ovf.ctr"}) c->bufs[n].per_mesh = 0;
  P->segs = (Seg*)c->bufs["ovf.segs"].p; P->ring = (double*)c->bufs["ovf.ring"].p; P->work = (unsigned char*)c->bufs["ovf.work"].p;
  P->seg_cap = c->ovf_seg_cap; P->ring_cap = c->ovf_ring_cap; P->work_cap = c->ovf_work_cap;
  P->ctr = (unsigned long long*)c->bufs["ovf.ctr"].p;
  return SH_OK;
}
// the plan arrays of slice set `pfx` (N planes per humerus), window-relative like every [B][...] buffer
static int ovf_set(sh_ctx* c, const std::string& pfx, int N, OvfSet* S) {
  const size_t B = (size_t)c->B;
  int rc;
  struct A { const char* suffix; size_t elem; } arr[6] = {{".ovf_soff", 8}, {".ovf_roff", 8}, {".ovf_woff", 8}, {".ovf_fill", 4}, {".ovf_list", 4}, {".ovf_list2", 4}};
  for (const A& a : arr) {
    const std::string nm = pfx + a.suffix;
    const bool fresh = c->bufs.find(nm) == c->bufs.end() || c->bufs[nm].bytes < B * N * a.elem;
    if ((rc = ensure(c, nm.c_str(), B * N * a.elem, (int)a.elem)) != SH_OK) return rc;
    c->bufs[nm].per_mesh = (size_t)N * a.elem;
    if (fresh) HIPCHK(c, hipMemsetAsync(c->bufs[nm].p, 0xFF, B * N * a.elem, c->stream));      // "no overflow plane" until a plan says otherwise
  }
  if ((rc = ensure(c, (pfx + ".ovf_nlist").c_str(), 16, 4)) != SH_OK) return rc;
  c->bufs[pfx + ".ovf_nlist"].per_mesh = 0;
  S->soff = buf<long long>(c, (pfx + ".ovf_soff").c_str()); S->roff = buf<long long>(c, (pfx + ".ovf_roff").c_str());
  S->woff = buf<long long>(c, (pfx + ".ovf_woff").c_str()); S->fill = buf<int>(c, (pfx + ".ovf_fill").c_str());
  S->list = buf<int>(c, (pfx + ".ovf_list").c_str()); S->nlist = (int*)c->bufs[pfx + ".ovf_nlist"].p; S->list2 = buf<int>(c, (pfx + ".ovf_list2").c_str());
  return SH_OK;
}

// One or two slice sets through the set's launches together (k_slices.h SliceSets): plane heights (+ counter clears, bound decode),
// one pass over the mesh for their sections, one grid for their joins; the overflow tier and the resampling stay per set.
struct SliceSpec { const char* pfx; int kind, N; bool ring, resample; int select; bool total_area, decode_bounds; };
static int run_slice_sets(sh_ctx* c, const SliceSpec* specs, int nspec) {
  const int B = c->Bwin;
  if (nspec < 1 || nspec > 2) return fail(c, SH_ERR_STATE, "run_slice_sets: one or two sets");
  int ntot = 0;
  for (int i = 0; i < nspec; ++i) ntot += specs[i].N;
  if (ntot > SH_EMIT_MAXN) return fail(c, SH_ERR_CAPACITY, "slice sets have more planes than k_slice_emit's LDS histogram");
  OvfPools OP; OvfSet OS[2];
  { int orc; if ((orc = ovf_pools(c, &OP)) != SH_OK) return orc; }
  const bool ovf_on = c->ovf_none_gen != c->batch_gen;      // (known from an earlier run of this batch: no plane overflows)
  SliceSets sets{};
  sets.n = nspec;
  sets.vobb = buf<double>(c, "verts_obb"); sets.voff = buf<long long>(c, "voff");
  for (int i = 0; i < nspec; ++i) {
    const SliceSpec& sp = specs[i];
    const std::string p = sp.pfx;
    { int orc; if ((orc = ovf_set(c, p, sp.N, &OS[i])) != SH_OK) return orc; }
    SliceSetDev& S = sets.s[i];
    S.N = sp.N; S.kind = sp.kind; S.select = sp.select;
    S.zb = buf<double>(c, sp.kind == 4 ? "obb.zb_pre" : "z_bounds");
    S.zs = buf<double>(c, (p + ".zs").c_str()); S.zeff = buf<double>(c, (p + ".zeff").c_str());
    S.seg_count = buf<int>(c, (p + ".seg_count").c_str()); S.segs = buf<Seg>(c, (p + ".segs").c_str());
    S.centroids = buf<double>(c, (p + ".centroids").c_str()); S.areas = buf<double>(c, (p + ".areas").c_str()); S.nloops = buf<int>(c, (p + ".nloops").c_str());
    S.ring_n = buf<int>(c, (p + ".ring_n").c_str());
    S.ring = sp.ring ? buf<double>(c, (p + ".ring").c_str()) : (double*)nullptr;
    S.areas_total = sp.total_area ? buf<double>(c, (p + ".area_total").c_str()) : (double*)nullptr;
    S.nlarge = (int*)c->bufs["slices.nlarge"].p + (sp.kind & 7);      // (one counter per kind of set)
    // planes with more than SH_MAXLOOPS loops: listed for the overflow tier's join (tier skipped: flagged, sh_collect runs again with it)
    S.many = ManyLoops{ovf_on ? OS[i].list2 : (int*)nullptr, ovf_on ? OS[i].nlist + 1 : (int*)nullptr, ovf_on ? (unsigned long long*)nullptr : OP.ctr + 6};
    S.ovf_missed = ovf_on ? (unsigned long long*)nullptr : OP.ctr + 6;
    // the plane-height launch also zeroes the set's crossing counters and its large-tier counter, resets the overflow tier's words
    // (segments / workspace used: per launch group, by its first set; ring points stay for the run) and, for the first set behind
    // k_transform_verts, decodes the z bounds (run_window)
    S.aux = PlaneAux{sp.decode_bounds ? (const unsigned long long*)buf<unsigned long long>(c, "zb_enc") : (const unsigned long long*)nullptr, buf<double>(c, "z_bounds"), S.seg_count, S.nlarge,
                     ovf_on ? OS[i].nlist : (int*)nullptr, i == 0 ? OP.ctr : (unsigned long long*)nullptr};
  }
  LAUNCH(c, "k_make_planes", k_make_planes, dim3(B, nspec), dim3(256), sets, (const double*)buf<double>(c, "neck_z"), B);
  dim3 g((unsigned)std::min<long long>((c->maxF + 255) / 256, 4096), (unsigned)B);
  LAUNCH(c, "k_slice_emit", k_slice_emit, g, dim3(256), buf<double>(c, "verts_obb"), buf<int>(c, "faces"), buf<long long>(c, "voff"), buf<long long>(c, "foff"), sets);
  if (ovf_on)      // planes with more crossings than slots (k_ovf.h): plan their pool ranges, section them again into the segment pool
    for (int i = 0; i < nspec; ++i) {
      const SliceSetDev& S = sets.s[i];
      LAUNCH(c, "k_ovf_plan", k_ovf_plan, dim3((unsigned)((B * S.N + 255) / 256)), dim3(256), S.N, B * S.N, (const int*)S.seg_count, OP, OS[i], buf<int>(c, "err"));
      LAUNCH(c, "k_slice_emit_ovf", k_slice_emit_ovf, g, dim3(256), buf<double>(c, "verts_obb"), buf<int>(c, "faces"), buf<long long>(c, "voff"),
             buf<long long>(c, "foff"), (const double*)S.zeff, S.N, OP, OS[i]);
    }
  // two capacity tiers share the grid (k_slices.h): the planes of the other tier exit at once
  LAUNCH(c, "k_slice_link", k_slice_link, dim3(B * ntot), dim3(SH_LINK_THREADS), sets, B, buf<int>(c, "err"));
  LAUNCH(c, "k_slice_link_large", k_slice_link_large, dim3(std::min(B * ntot, 512)), dim3(SH_LINK_THREADS), sets, B, buf<int>(c, "err"));
  for (int i = 0; i < nspec; ++i) {
    const SliceSpec& sp = specs[i];
    const SliceSetDev& S = sets.s[i];
    if (ovf_on) {
      LAUNCH(c, "k_ovf_plan_loops", k_ovf_plan_loops, dim3(16), dim3(256), S.N, (const int*)S.seg_count, (const Seg*)S.segs, OP, OS[i], buf<int>(c, "err"));
      LAUNCH(c, "k_slice_link_huge", k_slice_link_huge, dim3(64), dim3(SH_HUGE_THREADS), S.N, (const int*)S.seg_count, OP, OS[i], S.centroids, S.areas, S.nloops, S.ring_n,
             sp.ring ? 1 : 0, S.select, buf<int>(c, "err"), S.areas_total, (const double*)buf<double>(c, "verts_obb"), (const long long*)buf<long long>(c, "voff"), (const double*)S.zeff);
    }
    if (sp.resample) {
      RsWant want{c->keep_products ? 1 : 0, SH_ANP_ROW0, 0, 0};
      cutoff_range(SH_NPROX, c->params.groove_cutoff[0], c->params.groove_cutoff[1], &want.cs_lo, &want.cs_hi);
      if (c->b0 == 0) { c->rs_cs_lo = want.cs_lo; c->rs_cs_hi = want.cs_hi; c->rs_all = c->keep_products; c->rs_gen = c->batch_gen; }
      LAUNCH(c, "k_resample_polar", k_resample_polar, dim3(B * S.N), dim3(SH_RS_THREADS), S.N, SH_MPROX, S.ring_n, S.ring,
             S.centroids, buf<double>(c, "prox.ixy"), buf<double>(c, "prox.itr_start"), buf<double>(c, "prox.itr_centered_start"), (const long long*)OS[i].roff, want);
      LAUNCH(c, "k_resample_polar_large", k_resample_polar_large, dim3(std::min(B * S.N, 512)), dim3(SH_RS_THREADS), B * S.N, S.N, SH_MPROX, S.ring_n, S.ring,
             S.centroids, buf<double>(c, "prox.ixy"), buf<double>(c, "prox.itr_start"), buf<double>(c, "prox.itr_centered_start"), (const int*)S.nlarge, (const long long*)OS[i].roff, want);
      if (ovf_on) {
        LAUNCH(c, "k_resample_polar_huge", k_resample_polar_huge, dim3(64), dim3(SH_RS_THREADS), S.N, SH_MPROX, (const int*)S.ring_n, OP, OS[i],
               S.centroids, buf<double>(c, "prox.ixy"), buf<double>(c, "prox.itr_start"), buf<double>(c, "prox.itr_centered_start"), want);
      }
    }
  }
  return SH_OK;
}
static int run_slice_set(sh_ctx* c, const char* pfx, int kind, int N, bool ring, bool resample, int select = 0, bool total_area = false, bool decode_bounds = false) {
  const SliceSpec sp{pfx, kind, N, ring, resample, select, total_area, decode_bounds};
  return run_slice_sets(c, &sp, 1);
}

// The hull's input points.  Host-provided batch: the caller's vertices.  Device-generated batch: the prefilter
// (k_hullpre.h) drops the vertices strictly inside a 26-direction polytope on the device and only the rest comes back
// (39 % of a humerus, into pinned memory).  Callable from the background thread: no buffer-map access, no timers.
struct HullPre { const float* verts; const long long* voff; int* ext; double* planes; int* npl; float* kept; int* nkept; long long* koff; double* pval; int* pidx; int* pcnt; long long* poff; };
static HullPre hullpre_ptrs(sh_ctx* c) {      // calling thread only (buffer map)
  return HullPre{(const float*)c->bufs["verts"].p, (const long long*)c->bufs["voff"].p, (int*)c->bufs["hullpre.ext"].p, (double*)c->bufs["hullpre.planes"].p,
                 (int*)c->bufs["hullpre.npl"].p, (float*)c->bufs["hullpre.kept"].p, (int*)c->bufs["hullpre.nkept"].p, (long long*)c->bufs["hullpre.koff"].p,
                 (double*)c->bufs["hullpre.pval"].p, (int*)c->bufs["hullpre.pidx"].p, (int*)c->bufs["hullpre.pcnt"].p, (long long*)c->bufs["hullpre.poff"].p};
}
// the five launches of the device prefilter (k_hullpre.h): survivors of all B humeri compacted into hp.kept at hp.koff
static void launch_prefilter(const HullPre& hp, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_hullpre_extremes, dim3(SH_HP_PARTS, B), dim3(256), 0, st, hp.verts, hp.voff, hp.pval, hp.pidx);
  hipLaunchKernelGGL(k_hullpre_polytope, dim3(B), dim3(256), 0, st, hp.verts, hp.voff, (const double*)hp.pval, (const int*)hp.pidx, hp.ext, hp.planes, hp.npl);
  hipLaunchKernelGGL(k_hullpre_filter<false>, dim3(SH_HP_PARTS, B), dim3(256), 0, st, hp.verts, hp.voff, (const double*)hp.planes, (const int*)hp.npl,
                     (const long long*)hp.poff, hp.kept, hp.pcnt);
  hipLaunchKernelGGL(k_hullpre_offsets, dim3(1), dim3(64), 0, st, (const int*)hp.pcnt, hp.koff, hp.poff, hp.nkept, B);
  hipLaunchKernelGGL(k_hullpre_filter<true>, dim3(SH_HP_PARTS, B), dim3(256), 0, st, hp.verts, hp.voff, (const double*)hp.planes, (const int*)hp.npl,
                     (const long long*)hp.poff, hp.kept, hp.pcnt);
}

// survivors of the device prefilter of a batch of B humeri (sumV vertices in all) -> pinned h_kept / h_koff, described by *out
static hipError_t fetch_prefiltered(const HullPre& hp, int B, long long sumV, long long* h_koff, float* h_kept, sh_ctx::HullPts* out, hipStream_t st) {
  hipError_t e;
  launch_prefilter(hp, B, st);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(h_koff, hp.koff, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  const long long total = h_koff[B];
  if (total < 0 || total > sumV) return hipErrorUnknown;
  if (total > 0 && (e = hipMemcpyAsync(h_kept, hp.kept, (size_t)total * 12, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;      // one copy for the batch
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  out->off.resize(B); out->cnt.resize(B);
  for (int b = 0; b < B; ++b) { out->off[b] = h_koff[b]; out->cnt[b] = (int)(h_koff[b + 1] - h_koff[b]); }
  out->src = h_kept;
  return hipSuccess;
}

static hipError_t fetch_hull_points(sh_ctx* c, const HullPre& hp, hipStream_t st) {
  const int B = c->B;
  sh_ctx::HullPts& in = c->hull_in;
  in.cnt.resize(B);
  if (c->h_verts_valid) {
    in.off.assign(c->h_voff.begin(), c->h_voff.begin() + B);
    for (int b = 0; b < B; ++b) in.cnt[b] = (int)(c->h_voff[b + 1] - c->h_voff[b]);
    in.src = c->h_verts.data();
    return hipSuccess;
  }
  static const bool prefilter = !(getenv("SHOULDER_HULL_PREFILTER") && getenv("SHOULDER_HULL_PREFILTER")[0] == '0');
  hipError_t e;
  if (!prefilter || !hp.kept) {
    c->h_verts.resize(3 * (size_t)c->sumV);
    if ((e = hipMemcpyAsync(c->h_verts.data(), hp.verts, c->sumV * 3 * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    in.off.assign(c->h_voff.begin(), c->h_voff.begin() + B);
    for (int b = 0; b < B; ++b) in.cnt[b] = (int)(c->h_voff[b + 1] - c->h_voff[b]);
    in.src = c->h_verts.data();
    return hipSuccess;
  }
  return fetch_prefiltered(hp, B, c->sumV, c->h_koff, c->h_kept, &in, st);
}

// Process-wide worker pool of the host hull phase.  A job is a callable every participating thread runs to completion
// (the callable itself hands out mesh indices through an atomic counter); run() returns when all workers that picked
// the job up have left it.  Jobs of different contexts queue up FIFO and are served by the same threads.
class HullPool {
 public:
  static HullPool& instance() { static HullPool p; return p; }
  static unsigned thread_count() {
    unsigned nt = std::min(threads_per_local_rank(false), 32u);      // (affinity mask / LOCAL_WORLD_SIZE aware)
    if (const char* e = getenv("SHOULDER_HULL_THREADS")) { int v = atoi(e); if (v > 0) nt = (unsigned)v; }
    return nt;
  }
  void run(const std::function<void()>& fn, int items) {
    Job job;
    job.fn = fn;
    job.want = (int)std::min<unsigned>(std::max(1, items), (unsigned)workers_.size() + 1) - 1;      // helpers besides the caller
    if (job.want > 0) {
      { std::lock_guard<std::mutex> lk(mu_); queue_.push_back(&job); }
      cv_.notify_all();
    }
    fn();                                    // the caller takes part
    if (job.want > 0) {
      std::unique_lock<std::mutex> lk(mu_);
      // helpers that have not started yet are no longer needed (the counter inside fn is exhausted): withdraw the job
      auto it = std::find(queue_.begin(), queue_.end(), &job);
      if (it != queue_.end()) queue_.erase(it);
      done_cv_.wait(lk, [&] { return job.active == 0; });
    }
  }

 private:
  struct Job { std::function<void()> fn; int want = 0, taken = 0, active = 0; };
  HullPool() {
    const unsigned nt = thread_count();
    for (unsigned t = 1; t < nt; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~HullPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
      if (stop_) return;
      Job* j = queue_.front();
      ++j->taken; ++j->active;
      if (j->taken >= j->want) queue_.pop_front();
      lk.unlock();
      j->fn();
      lk.lock();
      if (--j->active == 0) done_cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<Job*> queue_;
  std::vector<std::thread> workers_;
  bool stop_ = false;
};

// mesh.py:63-125.  Host: one quickhull per humerus on worker threads (sh_hull.h).  Device: candidate
// boxes for every hull face, pick + frame, end sections, circle fits, flip (k_obb.h).
// Host phase of the OBB stage for meshes [b0, b0 + B): one quickhull per humerus on worker threads into pinned slot
// `slot`.  Callable from the background thread: touches no error string, no timers; HIP errors come back as text.
// One hull phase at a time per process, the one a run is WAITING for first.  The pool is shared by the lanes of a process; two
// phases at once (a run's own and another lane's background preparation) interleaved on the same workers and both came late:
// measured at the start of a timed region, lanes idle -- the second lane's first hull phase took 12.8 ms instead of 5.1 beside
// the first lane's preparation of its NEXT step, the first UNet passes were 10-17 ms apart and 20 steps carried 0.5-0.8 ms each
// of it.  A background preparation now waits while a foreground phase is running or waiting, and one that is under way hands the
// pool over at the next hull boundary (its workers take no further humerus; it finishes the rest after the foreground phase).
struct HullPhaseGate {
  std::mutex m; std::condition_variable cv; bool busy = false; std::atomic<int> fg_waiting{0};
  void enter(bool background) {
    std::unique_lock<std::mutex> l(m);
    if (!background) ++fg_waiting;
    cv.wait(l, [&] { return !busy && (!background || fg_waiting.load() == 0); });
    if (!background) --fg_waiting;
    busy = true;
  }
  bool foreground_waits() const { return fg_waiting.load(std::memory_order_relaxed) > 0; }
  void leave() { { std::lock_guard<std::mutex> l(m); busy = false; } cv.notify_all(); }
  static HullPhaseGate& instance() { static HullPhaseGate g; return g; }
};

static int hull_host_phase(sh_ctx* c, const sh_ctx::HullPts& in, int slot, int b0, int B, int* bad_mesh, double* ms, std::string* errtxt, bool background = false) {
  auto t0 = std::chrono::steady_clock::now();
  sh_ctx::HullStage& hs = c->hstage[slot];
#define HULLCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *errtxt = std::string(#call) + ": " + hipGetErrorString(e_); return SH_ERR_HIP; } } while (0)
  bool grown = false;
again:
  if (hs.cap < B || grown) {
    if (hs.hv) { (void)hipHostFree(hs.hv); (void)hipHostFree(hs.nr); (void)hipHostFree(hs.ed); (void)hipHostFree(hs.cnt); }
    hs.hv = nullptr; hs.cap = 0;
    HULLCHK(hipHostMalloc((void**)&hs.hv, (size_t)B * hs.pv * 3 * 8));
    HULLCHK(hipHostMalloc((void**)&hs.nr, (size_t)B * hs.pf * 3 * 8));
    HULLCHK(hipHostMalloc((void**)&hs.ed, (size_t)B * hs.pe * 4 * 4));
    HULLCHK(hipHostMalloc((void**)&hs.cnt, (size_t)B * 3 * 4));
    hs.cap = B;
  }
  if (!hs.ev) HULLCHK(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
  if (hs.used) HULLCHK(hipEventSynchronize(hs.ev));     // the previous copies out of this slot are done
  {
    // the other slot of the double buffer is allocated with the first one: pinning its 31 MB costs ~7 ms, and a context that had run
    // once (a warm-up step) paid that in its SECOND run -- the first timed step of a lane (round 3: 12 ms instead of 5 for that hull
    // phase, the first UNet passes of a 20-step region 14 ms apart)
    sh_ctx::HullStage& ho = c->hstage[slot ^ 1];
    if (!ho.hv && !grown) {
      ho.pv = hs.pv; ho.pf = hs.pf; ho.pe = hs.pe;
      HULLCHK(hipHostMalloc((void**)&ho.hv, (size_t)B * ho.pv * 3 * 8));
      HULLCHK(hipHostMalloc((void**)&ho.nr, (size_t)B * ho.pf * 3 * 8));
      HULLCHK(hipHostMalloc((void**)&ho.ed, (size_t)B * ho.pe * 4 * 4));
      HULLCHK(hipHostMalloc((void**)&ho.cnt, (size_t)B * 3 * 4));
      ho.cap = B;
    }
  }
#undef HULLCHK
  double* hv = hs.hv; double* nr = hs.nr; int* ed = hs.ed; int* counts = hs.cnt;
  std::vector<int> status(B, 0);
  std::vector<int> demand(3 * (size_t)B, 0);      // of the humeri whose hull does not fit the staging pitch
  static const bool gate_on = !(getenv("SHOULDER_HULL_GATE") && getenv("SHOULDER_HULL_GATE")[0] == '0');
  std::atomic<int> next(0);
  auto work = [&]() {
    std::vector<double> P;
    shhull::Hull H;
    for (;;) {
      if (background && gate_on && HullPhaseGate::instance().foreground_waits()) break;      // a run is waiting for ITS hulls: hand the pool over
      int b = next.fetch_add(1);
      if (b >= B) break;
      counts[b] = counts[B + b] = counts[2 * B + b] = 0;
      long long v0 = in.off[b0 + b], nv = in.cnt[b0 + b];
      P.resize(3 * (size_t)nv);
      const float* src = in.src + 3 * v0;
      for (long long i = 0; i < 3 * nv; ++i) P[i] = (double)src[i];
      if (!shhull::convex_hull(P.data(), (int)nv, H)) { status[b] = SH_ERR_GEOMETRY; continue; }
      int hn = (int)H.vert_ids.size(), fn = (int)H.tris.size() / 3, en = (int)H.edges.size() / 4;
      if (hn > hs.pv || fn > hs.pf || en > hs.pe) { status[b] = 1; demand[3 * (size_t)b] = hn; demand[3 * (size_t)b + 1] = fn; demand[3 * (size_t)b + 2] = en; continue; }      // does not fit the staging pitch: see below
      for (int i = 0; i < hn; ++i)
        for (int k = 0; k < 3; ++k) hv[((size_t)b * hs.pv + i) * 3 + k] = P[3 * (size_t)H.vert_ids[i] + k];
      std::copy(H.normals.begin(), H.normals.end(), nr + (size_t)b * hs.pf * 3);
      std::copy(H.edges.begin(), H.edges.end(), ed + (size_t)b * hs.pe * 4);
      counts[b] = hn; counts[B + b] = fn; counts[2 * B + b] = en;
    }
  };
  // One pool of worker threads per process, started once and shared by every context (lane) of the process: the host's
  // hardware threads divided between the ranks of this node (torchrun exports LOCAL_WORLD_SIZE), at most 32 per process;
  // SHOULDER_HULL_THREADS overrides.  The calling thread works on its own batch too.  (Round 1 started up to 32 threads
  // per batch: a third of the 4.8 ms hull phase was thread start-up, and two lanes doubled the thread count.)
  do {
    if (gate_on) HullPhaseGate::instance().enter(background);
    HullPool::instance().run(work, B);
    if (gate_on) HullPhaseGate::instance().leave();
  } while (next.load() < B);      // (a background phase that handed the pool over: the remaining humeri)
  if (!grown && std::find(status.begin(), status.end(), 1) != status.end()) {
    // a hull larger than the staging pitch (a dense mesh): this slot gets the record capacity -- or, above it, what the largest hull
    // of the batch needs (the device record grows at the upload: grow_hull_records) -- and the phase runs again
    int dv = SH_HV, df = SH_HF, de = SH_HE;
    for (int b = 0; b < B; ++b) { dv = std::max(dv, demand[3 * (size_t)b]); df = std::max(df, demand[3 * (size_t)b + 1]); de = std::max(de, demand[3 * (size_t)b + 2]); }
    auto up = [](int x) { return (x + 1023) / 1024 * 1024; };
    hs.pv = up(dv); hs.pf = up(df); hs.pe = up(de);
    grown = true;
    std::fill(status.begin(), status.end(), 0);
    next = 0;
    goto again;
  }
  *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (int b = 0; b < B; ++b)
    if (status[b] != 0) { *bad_mesh = b0 + b; return status[b] == 1 ? SH_ERR_CAPACITY : status[b]; }      // (1 survives only if the re-sized staging still did not hold a hull)
  return SH_OK;
}

// Hull records of pinned slot `slot` -> device buffers on stream `st`: only the used head of every fixed-capacity record
// crosses PCIe (one strided copy per array).  `dst` = {hull.hv, hull.normals, hull.edges, hull.nv, hull.nf, hull.ne}.
// do the hulls of pinned slot `slot` fit the device record?  (the background threads upload only when they do: growing re-allocates)
static bool hull_fits(const sh_ctx* c, int slot, int B, int* need /*[3] or null*/) {
  const int* counts = c->hstage[slot].cnt;
  int nvmax = 1, nfmax = 1, nemax = 1;
  for (int b = 0; b < B; ++b) { nvmax = std::max(nvmax, counts[b]); nfmax = std::max(nfmax, counts[B + b]); nemax = std::max(nemax, counts[2 * B + b]); }
  if (need) { need[0] = nvmax; need[1] = nfmax; need[2] = nemax; }
  return nvmax <= c->hcap.v && nfmax <= c->hcap.f && nemax <= c->hcap.e;
}
// A hull above the record's capacity (a strictly convex surface keeps every vertex: 16 384 is not a bound of the reference's
// `convex_hull`, mesh.py:82): the hull record and the per-face arrays of the OBB stage are re-allocated at strides that hold it.
// Foreground only, nothing of this context in flight reads them afterwards (hipFree waits for the device).
static int grow_hull_records(sh_ctx* c, int nv, int nf, int ne) {
  auto up = [](int x) { return (x + x / 8 + 1023) / 1024 * 1024; };
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (nv > c->hcap.v) c->hcap.v = up(nv);
  if (nf > c->hcap.f) c->hcap.f = up(nf);
  if (ne > c->hcap.e) c->hcap.e = up(ne);
  const size_t B = (size_t)c->B;
  const struct { const char* n; size_t per; int elem; } arr[] = {
      {"hull.hv", (size_t)c->hcap.v * 24, 8}, {"hull.normals", (size_t)c->hcap.f * 24, 8}, {"hull.edges", (size_t)c->hcap.e * 16, 4},
      {"obb.cand_vol", (size_t)c->hcap.f * 8, 8}, {"obb.cand_edge", (size_t)c->hcap.f * 4, 4}, {"obb.area2", (size_t)c->hcap.f * 8, 8},
      {"obb.lb", (size_t)c->hcap.f * 8, 8}, {"obb.dir_list", (size_t)c->hcap.f * 4, 4}, {"obb.seeded", (size_t)c->hcap.f, 1}};
  for (const auto& a : arr) {
    if (int e = ensure(c, a.n, B * a.per, a.elem)) return e;
    c->bufs[a.n].per_mesh = a.per;
  }
  if (c->sw.debug) fprintf(stderr, "[sh] hull record grown to %d vertices / %d faces / %d edges per humerus\n", c->hcap.v, c->hcap.f, c->hcap.e);
  return SH_OK;
}
static hipError_t hull_upload(sh_ctx* c, int slot, int B, void* const dst[6], hipStream_t st) {
  sh_ctx::HullStage& hs = c->hstage[slot];
  const int* counts = hs.cnt;
  int nvmax = 1, nfmax = 1, nemax = 1;
  for (int b = 0; b < B; ++b) { nvmax = std::max(nvmax, counts[b]); nfmax = std::max(nfmax, counts[B + b]); nemax = std::max(nemax, counts[2 * B + b]); }
  if (nvmax > c->hcap.v || nfmax > c->hcap.f || nemax > c->hcap.e) return hipErrorInvalidValue;      // (callers check hull_fits / grow first)
  hipError_t e;
  if ((e = hipMemcpy2DAsync(dst[0], (size_t)c->hcap.v * 24, hs.hv, (size_t)hs.pv * 24, (size_t)nvmax * 24, B, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpy2DAsync(dst[1], (size_t)c->hcap.f * 24, hs.nr, (size_t)hs.pf * 24, (size_t)nfmax * 24, B, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpy2DAsync(dst[2], (size_t)c->hcap.e * 16, hs.ed, (size_t)hs.pe * 16, (size_t)nemax * 16, B, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(dst[3], counts, (size_t)B * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(dst[4], counts + B, (size_t)B * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(dst[5], counts + 2 * B, (size_t)B * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipEventRecord(hs.ev, st)) != hipSuccess) return e;      // pinned slot is free again once these copies have run
  hs.used = true;
  return hipSuccess;
}

// mesh.py:63-125.  Host: convex hulls (hull_host_phase; already done by the background thread when `prepared_slot`
// >= 0).  Device: candidate boxes for every hull face, pick + frame, end sections, circle fits, flip (k_obb.h).
static bool device_hull_now(const sh_ctx* c) { return c->hull_mode == 1 && !(c->hull_force_host && c->obb_gen == c->batch_gen); }

static int run_obb(sh_ctx* c, int prepared_slot) {
  const int B = c->Bwin, b0 = c->b0;
  int nfmax = 1;
  if (c->obb_gen != c->batch_gen) { c->obb_gen = c->batch_gen; c->obb_sil_need = 0; c->obb_nf_over = false; c->hull_force_host = false; }
  if (c->redo_records) {
    nfmax = std::max(1, c->redo_nf);      // (redo_given_up put the host quickhull's records of this window into hull.*)
  } else if (device_hull_now(c)) {
    // hull on the device: prefilter -> round-based quickhull (k_hull.h), all on this context's stream; nothing comes to the host
    { int arc = alloc_hulld(c); if (arc != SH_OK) return arc; }
    launch_prefilter(hullpre_ptrs(c), B, c->stream);
    HIPCHK(c, hipGetLastError());
    HullScratch hs{buf<int>(c, "hulld.fv"), buf<int>(c, "hulld.vis"), buf<int>(c, "hulld.ev"), buf<int>(c, "hulld.hor"), buf<int>(c, "hulld.newslot"),
                   buf<int>(c, "hulld.freestack"), buf<unsigned long long>(c, "hulld.tkeys"), buf<unsigned>(c, "hulld.tvals")};
    LAUNCH(c, "k_hull_rounds", k_hull_rounds, dim3(B), dim3(HD_THREADS), (const float*)c->bufs["hullpre.kept"].p, (const long long*)c->bufs["hullpre.koff"].p, hs,
           buf<double>(c, "hull.hv"), buf<double>(c, "hull.normals"), buf<int>(c, "hull.edges"), buf<int>(c, "hull.nv"), buf<int>(c, "hull.nf"), buf<int>(c, "hull.ne"),
           buf<int>(c, "hulld.fail"), buf<int>(c, "hulld.rounds"), (const int*)buf<int>(c, "hulld.skip"), c->hcap);
    LAUNCH(c, "k_hull_flag", k_hull_flag, dim3((B + 63) / 64), dim3(64), buf<int>(c, "hulld.fail"), buf<int>(c, "err"), B);
    nfmax = std::max((int)HD_SLOTS, c->skip_nfmax);      // (the face counts stay on the device: the candidate kernel's tiles beyond a hull's faces return at once;
                                                         //  a humerus kept on the host hull may have more faces than the device hull has slots)
  } else {
  int slot = prepared_slot;
  if (slot < 0) {
    slot = c->hslot; c->hslot ^= 1;
    int bad = -1; double ms = 0; std::string et;
    int hrc = hull_host_phase(c, c->hull_in, slot, b0, B, &bad, &ms, &et);
    if (c->sw.debug) fprintf(stderr, "[sh] run_obb: foreground hull phase %.2f ms\n", ms);
    if (c->timing) { KTimer& h = c->timers["host.hull"]; h.ms += ms; h.n += 1; }
    if (hrc == SH_ERR_HIP) { c->err = et; return hrc; }
    if (hrc != SH_OK) { char m[96]; snprintf(m, sizeof m, "mesh %d: convex hull failed (%d)", bad, hrc); return fail(c, hrc, m); }
  }
  const int* counts = c->hstage[slot].cnt;
  for (int b = 0; b < B; ++b) nfmax = std::max(nfmax, counts[B + b]);
  {
    int need[3];
    if (!hull_fits(c, slot, B, need)) {      // (never with an early upload: the background threads upload only what fits)
      int grc = grow_hull_records(c, need[0], need[1], need[2]);
      if (grc != SH_OK) return grc;
    }
  }
  if (!(prepared_slot >= 0 && c->prep.uploaded)) {
    void* const dst[6] = {buf<double>(c, "hull.hv"), buf<double>(c, "hull.normals"), buf<int>(c, "hull.edges"), buf<int>(c, "hull.nv"), buf<int>(c, "hull.nf"), buf<int>(c, "hull.ne")};
    HIPCHK(c, hull_upload(c, slot, B, dst, c->stream));
  }
  }
  const int* cnt_nv = buf<int>(c, "hull.nv");
  const int* cnt_nf = buf<int>(c, "hull.nf");
  const int* cnt_ne = buf<int>(c, "hull.ne");
  {
    FILL(c, {buf<unsigned long long>(c, "obb.best_enc"), (size_t)B * 8, 0xFF} /*"no candidate volume yet"*/, {buf<unsigned long long>(c, "obb.lbmin_enc"), (size_t)B * 8, 0xFF},
         {buf<double>(c, "obb.area2"), (size_t)B * c->hcap.f * 8, 0}, {buf<int>(c, "obb.endcnt"), (size_t)B * 2 * 4, 0},
         {buf<unsigned long long>(c, "zb_enc"), (size_t)B * 16, 0xFF} /*z bounds: "nothing seen yet"*/, {buf<unsigned long long>(c, "anp.mm_enc"), (size_t)B * 16, 0xFF});
    c->bounds_cleared = true;
    const int nemax = 3 * nfmax / 2 + 3;      // (a closed triangulated surface: 2 E = 3 F)
    const HullCap hc = c->hcap;
    LAUNCH(c, "k_obb_face_area2", k_obb_face_area2, dim3((unsigned)((std::min(nemax, hc.e) + 255) / 256), (unsigned)B), dim3(256), buf<double>(c, "hull.hv"),
           buf<double>(c, "hull.normals"), buf<int>(c, "hull.edges"), cnt_ne, buf<double>(c, "obb.area2"), hc);
    const int bnd_tiles = (nfmax + SH_OBB_BND_DIRS - 1) / SH_OBB_BND_DIRS;
    LAUNCH(c, "k_obb_bounds", k_obb_bounds, dim3((unsigned)(bnd_tiles * ((B + 7) / 8) * 8)), dim3(SH_OBB_BND_THREADS),
           buf<double>(c, "hull.hv"), cnt_nv, buf<double>(c, "hull.normals"), cnt_nf, buf<double>(c, "obb.area2"), buf<double>(c, "obb.lb"),
           buf<unsigned long long>(c, "obb.lbmin_enc"), buf<double>(c, "obb.cand_vol"), buf<int>(c, "obb.cand_edge"), bnd_tiles, B, hc);
    // capacity tier of k_obb_candidates (k_obb.h): the small one unless a hull of this launch has more than 8 192 faces or a direction
    // of an earlier run of this batch had more than 512 silhouette edges; the workspace tier above 32 768 faces / 2 048 edges
    const bool huge = nfmax > 32768 || c->obb_sil_need > 2048 || c->obb_nf_over;
    const bool big = !huge && (nfmax > 8192 || c->obb_sil_need > 512);
    const int TT = (big || huge) ? 8 : SH_OBB_TILE;
    const int ntiles = (nfmax + TT - 1) / TT;
    ObbWs ws{};
    ws.need = (unsigned long long*)c->bufs["ovf.ctr"].p + 8;
    if (huge) {
      ws.nwg = 256;
      ws.silcap = std::max(hc.v + 64, c->obb_sil_need + c->obb_sil_need / 8);      // (a silhouette is a cycle of the hull's graph; more only on degenerate input: then the demand is recorded)
      int wrc;
      if ((wrc = ensure(c, "obb.ws_fmask", (size_t)ws.nwg * hc.f, 1)) != SH_OK || (wrc = ensure(c, "obb.ws_lists", (size_t)ws.nwg * 8 * ws.silcap * 4, 4)) != SH_OK ||
          (wrc = ensure(c, "obb.ws_sxy", (size_t)ws.nwg * ws.silcap * 16, 8)) != SH_OK || (wrc = ensure(c, "obb.ws_area", (size_t)ws.nwg * ws.silcap * 8, 8)) != SH_OK) return wrc;
      for (const char* n : {"obb.ws_fmask", "obb.ws_lists", "obb.ws_sxy", "obb.ws_area"}) c->bufs[n].per_mesh = 0;
      ws.fmask = (unsigned char*)c->bufs["obb.ws_fmask"].p; ws.lists = (unsigned*)c->bufs["obb.ws_lists"].p;
      ws.sxy = (double2*)c->bufs["obb.ws_sxy"].p; ws.area = (double*)c->bufs["obb.ws_area"].p;
    }
    // SHOULDER_OBB_PRUNE=0: every direction is evaluated (the A/B of the pruning bound: same frames, tests/test_gpu_hull.py)
    const bool prune = c->sw.obb_prune;
    for (int pass = 0; pass < 2; ++pass) {      // seed tile, then the directions its best volume cannot exclude
      LAUNCH(c, "k_obb_select", k_obb_select, dim3(B), dim3(256), buf<double>(c, "obb.lb"), cnt_nf, buf<unsigned long long>(c, "obb.lbmin_enc"),
             buf<unsigned long long>(c, "obb.best_enc"), (prune || pass == 0) ? pass : 2, buf<int>(c, "obb.dir_list"), buf<int>(c, "obb.dir_count"), buf<unsigned char>(c, "obb.seeded"), hc);
      const int nt_pass = pass == 0 ? SH_OBB_TILE / TT : ntiles;      // (the seed pass: up to SH_OBB_TILE directions)
      const dim3 cg((unsigned)(nt_pass * ((B + 7) / 8) * 8));
      if (huge) {
        LAUNCH(c, pass == 0 ? "k_obb_seed" : "k_obb_candidates", (k_obb_candidates<8, 1, 0, 0, unsigned, true>), dim3((unsigned)std::min<unsigned>(cg.x, (unsigned)ws.nwg)), dim3(SH_OBB_THREADS),
               buf<double>(c, "hull.hv"), cnt_nv, buf<double>(c, "hull.normals"), cnt_nf, buf<int>(c, "hull.edges"), cnt_ne, buf<double>(c, "obb.cand_vol"),
               buf<int>(c, "obb.cand_edge"), buf<int>(c, "err"), buf<unsigned long long>(c, "obb.best_enc"), buf<int>(c, "obb.dir_list"), buf<int>(c, "obb.dir_count"),
               nt_pass, B, prune ? 1 : 0, hc, ws);
      } else if (big) {
        LAUNCH(c, pass == 0 ? "k_obb_seed" : "k_obb_candidates", (k_obb_candidates<8, 1, 2048, 32768, unsigned>), cg, dim3(SH_OBB_THREADS),
               buf<double>(c, "hull.hv"), cnt_nv, buf<double>(c, "hull.normals"), cnt_nf, buf<int>(c, "hull.edges"), cnt_ne, buf<double>(c, "obb.cand_vol"),
               buf<int>(c, "obb.cand_edge"), buf<int>(c, "err"), buf<unsigned long long>(c, "obb.best_enc"), buf<int>(c, "obb.dir_list"), buf<int>(c, "obb.dir_count"),
               nt_pass, B, prune ? 1 : 0, hc, ws);
      } else {
        LAUNCH(c, pass == 0 ? "k_obb_seed" : "k_obb_candidates", (k_obb_candidates<SH_OBB_TILE, SH_OBB_GROUP, 512, 8192, unsigned short>), cg, dim3(SH_OBB_THREADS),
               buf<double>(c, "hull.hv"), cnt_nv, buf<double>(c, "hull.normals"), cnt_nf, buf<int>(c, "hull.edges"), cnt_ne, buf<double>(c, "obb.cand_vol"),
               buf<int>(c, "obb.cand_edge"), buf<int>(c, "err"), buf<unsigned long long>(c, "obb.best_enc"), buf<int>(c, "obb.dir_list"), buf<int>(c, "obb.dir_count"),
               nt_pass, B, prune ? 1 : 0, hc, ws);
      }
    }
  }
  LAUNCH(c, "k_obb_pick", k_obb_pick, dim3(B), dim3(256), buf<double>(c, "hull.hv"), buf<double>(c, "hull.normals"), cnt_nf, buf<int>(c, "hull.edges"),
         buf<double>(c, "obb.cand_vol"), buf<int>(c, "obb.cand_edge"), buf<float>(c, "verts"), buf<long long>(c, "voff"), buf<double>(c, "obb.T_pre"),
         buf<double>(c, "obb.zb_pre"), buf<int>(c, "err"), c->hcap);
  if (!c->obb_done_ev) HIPCHK(c, hipEventCreateWithFlags(&c->obb_done_ev, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->obb_done_ev, c->stream));      // hull.* device buffers are free for the next run's records from here
  if (c->params.bone_kind == SH_BONE_PROXIMAL) {
    // mesh.py:134-192 ProxObb: 100 sections of the mesh in the raw box frame, head = largest area, canal range
    dim3 gv((unsigned)std::min<long long>((c->maxV + 255) / 256, 1024), (unsigned)B);
    LAUNCH(c, "k_transform_verts", k_transform_verts, gv, dim3(256), buf<float>(c, "verts"), buf<long long>(c, "voff"),
           buf<double>(c, "obb.T_pre"), buf<double>(c, "verts_obb"), buf<unsigned long long>(c, "zb_enc"));      // (zb_enc: cleared by the fill above; its values are not used here)
    c->bounds_cleared = false;      // ... and run_window clears it again for the box frame's pass
    int rc2;
    if ((rc2 = run_slice_set(c, "pobb", 4, SH_NPSCAN, false, false, 0, true)) != SH_OK) return rc2;
    LAUNCH(c, "k_prox_obb", k_prox_obb, dim3((B + 63) / 64), dim3(64), buf<double>(c, "pobb.area_total"), buf<double>(c, "pobb.zs"), buf<double>(c, "obb.T_pre"),
           buf<double>(c, "obb_transform"), buf<int>(c, "flipped"), buf<double>(c, "pobb.cutoff"), buf<int>(c, "pobb.cutoff_idx"), buf<int>(c, "err"), B);
    c->obb_injected = true;
    return SH_OK;
  }
  dim3 g((unsigned)std::min<long long>((c->maxF + 255) / 256, 1024), (unsigned)B);
  LAUNCH(c, "k_obb_end_points", k_obb_end_points, g, dim3(256), buf<float>(c, "verts"), buf<int>(c, "faces"), buf<long long>(c, "voff"),
         buf<long long>(c, "foff"), buf<double>(c, "obb.T_pre"), buf<double>(c, "obb.zb_pre"), buf<double>(c, "obb.endpts"), buf<int>(c, "obb.endcnt"), c->end_cap);
  LAUNCH(c, "k_obb_ends", k_obb_ends, dim3(B), dim3(128), buf<double>(c, "obb.endpts"), buf<int>(c, "obb.endcnt"),
         buf<double>(c, "obb.T_pre"), buf<double>(c, "obb.resid"), buf<double>(c, "obb_transform"), buf<int>(c, "flipped"), buf<int>(c, "err"), B, c->end_cap,
         (unsigned long long*)c->bufs["ovf.ctr"].p + 7);
  c->obb_injected = true;
  return SH_OK;
}

// epicondyle.py:33-89, everything that needs the distal set only: the minimum-area rectangle of every distal slice in the cut (three
// capacity tiers) and the two ends of the widest one
static int run_te_rows(sh_ctx* c) {
  const int B = c->Bwin;
  OvfPools OP; OvfSet OS;
  { int orc; if ((orc = ovf_pools(c, &OP)) != SH_OK || (orc = ovf_set(c, "distal", SH_NDIST, &OS)) != SH_OK) return orc; }
  LAUNCH(c, "k_te_rows", k_te_rows<SH_SMALLSEG>, dim3(B * SH_TE_NROWS), dim3(64), buf<double>(c, "distal.ring"), buf<int>(c, "distal.ring_n"),
         buf<double>(c, "te.rects"), B, (const long long*)OS.roff);
  LAUNCH(c, "k_te_rows_large", k_te_rows<SH_MAXSEG>, dim3(B * SH_TE_NROWS), dim3(64), buf<double>(c, "distal.ring"), buf<int>(c, "distal.ring_n"),
         buf<double>(c, "te.rects"), B, (const long long*)OS.roff);
  if (c->ovf_none_gen != c->batch_gen) {
    LAUNCH(c, "k_te_rows_huge", k_te_rows_huge, dim3(64), dim3(64), OP, OS, (const int*)buf<int>(c, "distal.ring_n"), buf<double>(c, "te.rects"));
  }
  // epicondyle.py:39-89: the widest slice's end slivers, their centroids, the farthest pair (CT coordinates, piece order)
  LAUNCH(c, "k_te_ends", k_te_ends, dim3(B), dim3(64), buf<double>(c, "distal.ring"), buf<int>(c, "distal.ring_n"), buf<double>(c, "te.rects"),
         buf<double>(c, "distal.zs"), buf<double>(c, "obb_transform"), buf<double>(c, "te.ends_ct"), buf<int>(c, "te.row"), buf<int>(c, "err"), B, OP, OS);
  return SH_OK;
}

// All stages for the window [c->b0, c->b0 + c->Bwin) of the batch; everything is enqueued on the stream,
// nothing here waits for the device.
static int run_window(sh_ctx* c, uint32_t mask, int prepared_slot) {
  const int B = c->Bwin;
  int rc;
  c->bounds_cleared = false;
  if (mask & SH_STAGE_OBB)
    if ((rc = run_obb(c, prepared_slot)) != SH_OK) return rc;
  // the encoded minima / maxima (z bounds of the box frame, the anatomic-neck image's range) start as all ones: one fill for both,
  // in run_obb's first fill when that stage runs (they were a launch each)
  if (!c->bounds_cleared && (mask & (SH_STAGE_OBB | SH_STAGE_FULL | SH_STAGE_ANP)))
    FILL(c, {buf<unsigned long long>(c, "zb_enc"), (size_t)B * 16, 0xFF}, {buf<unsigned long long>(c, "anp.mm_enc"), (size_t)B * 16, 0xFF});
  const bool transformed = (mask & (SH_STAGE_OBB | SH_STAGE_FULL)) != 0;
  if (transformed) {
    // verts_obb + z bounds (mesh.py:85-86)
    dim3 g((unsigned)std::min<long long>((c->maxV + 255) / 256, 1024), (unsigned)B);
    LAUNCH(c, "k_transform_verts", k_transform_verts, g, dim3(256), buf<float>(c, "verts"), buf<long long>(c, "voff"),
           buf<double>(c, "obb_transform"), buf<double>(c, "verts_obb"), buf<unsigned long long>(c, "zb_enc"));
    if (!(mask & SH_STAGE_FULL))      // (with the full set in the run its k_make_planes decodes them)
      LAUNCH(c, "k_decode_bounds", k_decode_bounds, dim3((2 * B + 63) / 64), dim3(64), buf<unsigned long long>(c, "zb_enc"), buf<double>(c, "z_bounds"), B);
  }
  // Slice sets that hang on the same inputs share their launches (run_slice_sets): full + distal behind the box frame, neck contour +
  // proximal behind neck_z -- 8 launches and two passes over the mesh less per step; same sections (SHOULDER_SLICE_MERGE=0: one
  // set per launch group, the A/B of tests/test_gpu_slices.py)
  const bool merge_env = c->sw.slice_merge;
  // The distal set and the first part of the trans-epicondylar stage (the rectangles of its rows, the ends of the widest one) need
  // nothing but the box frame.  Small batches (up to 16 humeri: one humerus gains 4 %, 6.01 -> 5.78 ms per run; at B = 64 two streams'
  // kernels just share the CUs and one lane LOSES 8 %): the whole branch runs on the side stream beside the full -> neck -> canal ->
  // proximal -> groove chain.  Larger batches: the distal set stays in the chain; SHOULDER_TE_EARLY=1 forks only the
  // trans-epicondylar part (one lane's walk per slice: 0.24 + 0.32 ms of latency) so that it runs beside the lane's own UNet pass
  // instead of behind it and only k_te_orient (medial end first: needs the head's central axis) stays on the critical path.
  // Measured on the two-lane headline: 8.58 against 8.35 ms per step -- the 2 368 one-wave workgroups hold the 32 CUs the UNet
  // leaves free while the lane's own chain wants them -- so the fork is off by default (same records bit for bit either way) ...
  // Either fork only when the overflow tier is known to be idle for this batch (its pool counters are per set) and no per-launch
  // timing is on.
  const bool side_env = c->sw.side_stream;
  const bool te_early_env = c->sw.te_early == 1;
  const bool can_fork = side_env && (mask & SH_STAGE_DISTAL) && c->ovf_none_gen == c->batch_gen && c->timing != 1 && !c->redo_records;
  const bool side = can_fork && B <= 16;
  const bool te_early = can_fork && !side && te_early_env && (mask & SH_STAGE_TE) && (mask & SH_STAGE_ANP);
  bool te_rows_done = false;
  c->side_pending = false;
  const bool merge_fd = merge_env && (mask & SH_STAGE_FULL) && (mask & SH_STAGE_DISTAL) && !side;
  if (merge_fd) {
    // (both sets decode the z bounds themselves: their plane heights are made by ONE launch, the distal workgroups cannot wait for the
    //  full set's to write "z_bounds")
    const SliceSpec sp[2] = {{"full", 0, SH_NFULL, false, false, 0, false, transformed}, {"distal", 2, SH_NDIST, true, false, 0, false, transformed}};
    if ((rc = run_slice_sets(c, sp, 2)) != SH_OK) return rc;
  } else if (mask & SH_STAGE_FULL) {
    if ((rc = run_slice_set(c, "full", 0, SH_NFULL, false, false, 0, false, transformed)) != SH_OK) return rc;
  }
  if (mask & SH_STAGE_DISTAL) {
    hipStream_t main_stream = c->stream;
    auto fork = [&]() -> int {
      if (!c->side_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
      if (!c->side_fork_ev) { HIPCHK(c, hipEventCreateWithFlags(&c->side_fork_ev, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->side_join_ev, hipEventDisableTiming)); }
      HIPCHK(c, hipEventRecord(c->side_fork_ev, main_stream));
      HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->side_fork_ev, 0));
      c->stream = c->side_stream;
      return SH_OK;
    };
    if (side && (rc = fork()) != SH_OK) return rc;
    rc = merge_fd ? SH_OK : run_slice_set(c, "distal", 2, SH_NDIST, true, false);
    if (rc == SH_OK && te_early) rc = fork();
    // ... and by default it simply runs HERE, in the chain in front of the UNet pass instead of behind it: the same kernels on the same
    // stream, but the part of the step that follows the UNet -- what stands between the pass and the lane's next step -- is 0.3 ms
    // (0.6 ms beside the other lane's UNet) shorter: 8.00 -> 7.74 ms per step sustained, 8.48 -> 8.35 at 20 steps.  SHOULDER_TE_EARLY=0: behind the UNet.
    const bool te_inline = !side && !te_early && c->sw.te_early != 0 && (mask & SH_STAGE_ANP);
    if (rc == SH_OK && (side || te_early || te_inline) && (mask & SH_STAGE_TE)) { rc = run_te_rows(c); te_rows_done = rc == SH_OK; }
    if (side || te_early) {
      const bool forked = c->stream == c->side_stream;
      c->stream = main_stream;
      if (rc == SH_OK && forked) { HIPCHK(c, hipEventRecord(c->side_join_ev, c->side_stream)); c->side_pending = true; }
    }
    if (rc != SH_OK) return rc;
  }
  if (mask & SH_STAGE_NECK) {
    const bool prox = c->params.bone_kind == SH_BONE_PROXIMAL;      // surgical_neck.py:25-28
    if (prox) {
      LAUNCH(c, "k_neck", k_neck<true>, dim3(B), dim3(64), buf<double>(c, "full.areas"), buf<double>(c, "full.zs"),
             buf<double>(c, "neck_z"), buf<int>(c, "neck_index"), B, 0.2, 0.99, buf<double>(c, "neck.gram"));
    } else {
      LAUNCH(c, "k_neck", k_neck<false>, dim3(B), dim3(64), buf<double>(c, "full.areas"), buf<double>(c, "full.zs"),
             buf<double>(c, "neck_z"), buf<int>(c, "neck_index"), B, 0.70, 0.99, (double*)nullptr);
    }
    // surgical_neck.py:37-54: the contour at neck_z (loop whose vertex mean is nearest the origin) -- with the proximal set (which
    // starts from neck_z as well) in the same launches when both stages run
    if (merge_env && (mask & SH_STAGE_PROXIMAL)) {
      const SliceSpec sp[2] = {{"neckc", 3, 1, true, false, 1, false, false}, {"prox", 1, SH_NPROX, true, true, 0, false, false}};
      if ((rc = run_slice_sets(c, sp, 2)) != SH_OK) return rc;
    } else if ((rc = run_slice_set(c, "neckc", 3, 1, true, false, 1)) != SH_OK) return rc;
  }
  if (mask & SH_STAGE_CANAL) {
    LAUNCH(c, "k_canal", k_canal, dim3(B), dim3(64), buf<double>(c, "full.centroids"), buf<double>(c, "full.zs"),
           buf<double>(c, "z_bounds"), buf<double>(c, "obb_transform"), c->params.canal_cutoff[0], c->params.canal_cutoff[1],
           c->params.bone_kind == SH_BONE_PROXIMAL ? buf<double>(c, "pobb.cutoff") : (const double*)nullptr,
           buf<double>(c, "canal.points_obb"), buf<double>(c, "canal.axis_obb"), buf<double>(c, "canal.axis_ct"), buf<int>(c, "err"));
  }
  if ((mask & SH_STAGE_PROXIMAL) && !(merge_env && (mask & SH_STAGE_NECK)))
    if ((rc = run_slice_set(c, "prox", 1, SH_NPROX, true, true)) != SH_OK) return rc;
  if (mask & SH_STAGE_GROOVE) {
    if (!c->have_rfc) return fail(c, SH_ERR_STATE, "sh_run: groove stage needs sh_load_rfc first");
    int ga, gb;
    cutoff_range(SH_NPROX, c->params.groove_cutoff[0], c->params.groove_cutoff[1], &ga, &gb);
    // the centred polar rows [ga, gb) come from the proximal set's resampling, which writes the rows of ITS run's cut-off range (RsWant)
    if (!(mask & SH_STAGE_PROXIMAL) && !(c->rs_gen == c->batch_gen && (c->rs_all || (ga >= c->rs_cs_lo && gb <= c->rs_cs_hi))))
      return fail(c, SH_ERR_STATE, "sh_run: SH_STAGE_GROOVE without SH_STAGE_PROXIMAL, and the proximal slices of this batch were not made for this groove_cutoff (run SH_STAGE_PROXIMAL again)");
    const char* pp = buf<char>(c, "params") + c->unet_floats * 4;
    const size_t N = c->h_feat.size();
    const int* feat = (const int*)pp; const float* thr = (const float*)(pp + N * 4); const int* ti = (const int*)(pp + N * 8);
    const int* fi = (const int*)(pp + N * 12); const float* lw = (const float*)(pp + N * 16); const int* roots = (const int*)(pp + N * 20);
    const int rows = B * SH_GROOVE_NROWS;
    LAUNCH(c, "k_groove_rows", k_groove_rows, dim3(rows), dim3(64), buf<double>(c, "prox.itr_centered_start"),
           buf<double>(c, "prox.zs"), buf<double>(c, "canal.axis_ct"), ga, buf<double>(c, "groove.xraw"),
           buf<double>(c, "groove.ptheta"), buf<int>(c, "groove.npk"), buf<double>(c, "groove.r0"), buf<int>(c, "err"), B);
    LAUNCH(c, "k_groove_scale", k_groove_scale, dim3(B), dim3(256), buf<double>(c, "groove.xraw"), buf<int>(c, "groove.npk"),
           buf<double>(c, "groove.stats"), B, buf<int>(c, "groove.slots"), buf<int>(c, "groove.nslot"), buf<float>(c, "groove.proba"));
    if ((rc = ensure(c, "rfc.nodes", N * 16, 4)) != SH_OK) return rc;
    int4* nodes = (int4*)c->bufs["rfc.nodes"].p;
    if (!c->packed_rfc) {      // once per parameter block (it was a launch of every step's chain: 5 us alone, ~50 us beside a UNet pass)
      LAUNCH(c, "k_rfc_pack", k_rfc_pack, dim3((unsigned)((N + 255) / 256)), dim3(256), feat, thr, ti, fi, lw, nodes, (int)N);
      c->packed_rfc = true;
    }
    LAUNCH(c, "k_groove_rfc", k_groove_rfc, dim3((unsigned)(B * ((SH_GSLOTS + 63) / 64))), dim3(64), buf<double>(c, "groove.xraw"), buf<int>(c, "groove.slots"), buf<int>(c, "groove.nslot"),
           buf<double>(c, "groove.stats"), nodes, roots, c->rfc_trees, buf<double>(c, "groove.xs"), buf<float>(c, "groove.proba"), B);
    LAUNCH(c, "k_groove_tail", k_groove_tail, dim3(B), dim3(256), buf<double>(c, "groove.ptheta"), buf<float>(c, "groove.proba"), buf<double>(c, "groove.bg_theta"), buf<int>(c, "err"),
           buf<double>(c, "prox.itr_centered_start"), buf<double>(c, "groove.r0"), buf<double>(c, "prox.zs"), buf<double>(c, "prox.centroids"), ga, c->params.groove_deg_window,
           buf<int>(c, "groove.local_idx"), buf<double>(c, "groove.points_obb"), buf<double>(c, "obb_transform"), buf<double>(c, "groove.axis_ct"), buf<double>(c, "groove.points_ct"));
  }
  if (mask & SH_STAGE_ANP) {
    if (!c->have_unet) return fail(c, SH_ERR_STATE, "sh_run: anatomic-neck stage needs sh_load_unet first");
    LAUNCH(c, "k_anp_rows", k_anp_rows, dim3(B * SH_ANP_ROWS), dim3(64), buf<double>(c, "prox.itr_start"),
           buf<double>(c, "groove.bg_theta"), buf<double>(c, "anp.raw"), buf<double>(c, "anp.t01"), buf<int>(c, "anp.roll"), B,
           buf<unsigned long long>(c, "anp.mm_enc"));      // (+ the image's minimum / maximum: no second pass over it)
    // MinMaxScaler (anatomic_neck.py:56-58): the 16-bit network's first kernel applies it where it reads its patches (k_unet16_l0.h) --
    // no f32 image, 201 MB less traffic and a launch less per step; the other forms of the network and sh_set_keep_products get "anp.image"
    const bool scale_in_net = (c->params.unet_dtype == SH_UNET_BF16 || c->params.unet_dtype == SH_UNET_F16) && unet16_starts_fused(c, SH_ANP_ROWS, SH_MPROX);
    if (!scale_in_net || c->keep_products)
      LAUNCH(c, "k_anp_scale", k_anp_scale, dim3(64, B), dim3(256), buf<double>(c, "anp.raw"), buf<unsigned long long>(c, "anp.mm_enc"), buf<float>(c, "anp.image"));
    if ((rc = unet_turn_enter(c)) != SH_OK) return rc;
    if (scale_in_net) { c->unet_raw = buf<double>(c, "anp.raw"); c->unet_mm = buf<unsigned long long>(c, "anp.mm_enc"); }
    rc = unet_dispatch(c, buf<float>(c, "anp.image"), buf<float>(c, "anp.logits"), B, SH_ANP_ROWS, SH_MPROX);
    c->unet_raw = nullptr; c->unet_mm = nullptr;
    (void)unet_turn_leave(c);      // also after a failed pass: whatever was enqueued is what the next context waits for
    if (rc != SH_OK) return rc;
    LAUNCH(c, "k_anp_edge_count", k_anp_edge_count, dim3(SH_ANP_ROWS / 8, B), dim3(512), buf<float>(c, "anp.logits"), buf<int>(c, "anp.rowcnt"), buf<unsigned long long>(c, "anp.maskbits"));
    LAUNCH(c, "k_anp_edges", k_anp_edges, dim3(SH_ANP_ROWS / 8, B), dim3(512), buf<unsigned long long>(c, "anp.maskbits"), buf<double>(c, "anp.raw"),
           buf<double>(c, "anp.t01"), buf<int>(c, "anp.roll"), buf<double>(c, "prox.zs"), buf<int>(c, "anp.rowcnt"), buf<double>(c, "anp.points_obb"), buf<int>(c, "anp.counts"),
           buf<int>(c, "err"));
    LAUNCH(c, "k_anp_plane", k_anp_plane, dim3(B), dim3(256), buf<double>(c, "anp.points_obb"), buf<int>(c, "anp.counts"),
           buf<double>(c, "anp.plane"), buf<int>(c, "err"), buf<unsigned long long>(c, "anp.ray_t"));      // (+ "no hit yet" for the rays)
    LAUNCH(c, "k_rays_hit", k_rays_hit, dim3(SH_RAY_CHUNKS, B), dim3(256), buf<double>(c, "verts_obb"), buf<int>(c, "faces"), buf<long long>(c, "voff"),
           buf<long long>(c, "foff"), buf<double>(c, "anp.plane"), buf<unsigned long long>(c, "anp.ray_t"));
  }
  if (mask & SH_STAGE_TE) {
    if (c->side_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->side_join_ev, 0)); c->side_pending = false; }      // (rectangles and ends: done on the side stream)
    if (!te_rows_done && (rc = run_te_rows(c)) != SH_OK) return rc;
  }
  if (c->side_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->side_join_ev, 0)); c->side_pending = false; }
  // metrics of bone_props.py (side, retroversion, neck-shaft angle, radius of curvature): the sphere's sums need the mask and the neck plane only
  const uint32_t need = SH_STAGE_GROOVE | SH_STAGE_ANP | SH_STAGE_CSYS | (c->params.bone_kind == SH_BONE_PROXIMAL ? 0u : (uint32_t)SH_STAGE_TE);
  const bool metrics = (mask & need) == need;
  if (metrics)
    LAUNCH(c, "k_sphere_partial", k_sphere_partial, dim3(SH_SPH_PARTS, B), dim3(256), buf<unsigned long long>(c, "anp.maskbits"), buf<double>(c, "anp.raw"),
           buf<double>(c, "anp.t01"), buf<int>(c, "anp.roll"), buf<double>(c, "prox.zs"), buf<double>(c, "anp.plane"), buf<double>(c, "metrics.partial"));
  {      // ray points -> trans-epicondylar order -> record -> metrics: one launch, one workgroup per humerus (k_tail, k_te.h)
    PackArgs A{};
    A.lm = buf<sh_landmarks>(c, "landmarks"); A.T_obb = buf<double>(c, "obb_transform"); A.zb = buf<double>(c, "z_bounds"); A.neck_z = buf<double>(c, "neck_z");
    A.neck_index = buf<int>(c, "neck_index"); A.flipped = buf<int>(c, "flipped"); A.canal_axis_ct = buf<double>(c, "canal.axis_ct"); A.te_axis_ct = buf<double>(c, "te.axis_ct");
    A.groove_axis_ct = buf<double>(c, "groove.axis_ct"); A.bg_theta = buf<double>(c, "groove.bg_theta"); A.groove_pts_ct = buf<double>(c, "groove.points_ct");
    A.plane = buf<double>(c, "anp.plane"); A.axes_obb = buf<double>(c, "anp.axes_obb"); A.anp_pts_obb = buf<double>(c, "anp.points_obb"); A.anp_counts = buf<int>(c, "anp.counts");
    A.err = buf<int>(c, "err"); A.mask = mask; A.B = B; A.bone_kind = (int)c->params.bone_kind;
    A.canal_cut = c->params.bone_kind == SH_BONE_PROXIMAL ? buf<double>(c, "pobb.cutoff") : (const double*)nullptr;
    A.cc0 = c->params.canal_cutoff[0]; A.cc1 = c->params.canal_cutoff[1];
    A.ray_t = buf<unsigned long long>(c, "anp.ray_t"); A.te_ends_ct = buf<double>(c, "te.ends_ct"); A.sphere_partial = buf<double>(c, "metrics.partial"); A.metrics = metrics ? 1 : 0;
    LAUNCH(c, "k_tail", k_tail, dim3(B), dim3(256), A);
  }
  if (mask & SH_STAGE_APPLY) {      // bone.py:155: the mesh of every humerus in its own canal / trans-epicondylar (or canal / articular) frame
    dim3 g((unsigned)std::min<long long>((c->maxV + 255) / 256, 1024), (unsigned)B);
    LAUNCH(c, "k_apply_csys", k_apply_csys, g, dim3(256), buf<sh_landmarks>(c, "landmarks"), buf<float>(c, "verts"), buf<long long>(c, "voff"), buf<double>(c, "verts_csys"));
  }
  return SH_OK;
}

// Humeri per window.  Default: one window = the whole batch.  Measured on MI355X (bf16 UNet): B=256 as 4 windows of 64
// 2549 humeri/s vs 2744 as one window -- the kernels run ~15 % more efficiently on the larger launches, which is more
// than the hidden host hull (13 % of the step) buys back.  SHOULDER_WINDOW=<n> enables windows of n.
#define SH_WINDOW (1 << 30)

// ---- overlap of the host hulls with the device work of the previous run ----------------------------------
static int join_prepared(sh_ctx* c) {
  if (!c->prep.active) return SH_OK;
  if (c->prep.th.joinable()) c->prep.th.join();
  c->prep.active = false;
  if (c->timing) {
    KTimer& h = c->timers["host.hull"]; h.ms += c->prep.hull_ms; h.n += 1;
    if (c->prep.d2h_ms > 0) { KTimer& a = c->timers["host.verts_d2h"]; a.ms += c->prep.d2h_ms; a.n += 1; }
  }
  return c->prep.rc;      // a failed preparation is simply not used: the run repeats the host phase and reports the error itself
}

static void start_prepare(sh_ctx* c) {
  sh_ctx::Prepared& p = c->prep;
  p.active = true; p.slot = c->hslot; p.B = c->B; p.gen = c->batch_gen; p.rc = SH_OK; p.bad_mesh = -1; p.d2h_ms = p.hull_ms = 0; p.err.clear();
  p.uploaded = false; p.staged = false;
  // device pointers are looked up here: the buffer map belongs to the calling thread
  const HullPre hp = hullpre_ptrs(c);
  struct Dst { void* p[6]; } dst = {{buf<double>(c, "hull.hv"), buf<double>(c, "hull.normals"), buf<int>(c, "hull.edges"), buf<int>(c, "hull.nv"), buf<int>(c, "hull.nf"), buf<int>(c, "hull.ne")}};
  const bool can_upload = c->obb_done_ev != nullptr;
  p.th = std::thread([c, hp, dst, can_upload]() {
    sh_ctx::Prepared& q = c->prep;
    if (hipSetDevice(c->device) != hipSuccess) { q.rc = SH_ERR_HIP; return; }
    if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { q.rc = SH_ERR_HIP; return; }
    {      // the points of a device-generated batch come back for every run, on the copy stream, beside the kernels
      auto t0 = std::chrono::steady_clock::now();
      if (fetch_hull_points(c, hp, c->copy_stream) != hipSuccess) { q.rc = SH_ERR_HIP; return; }
      q.d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    q.rc = hull_host_phase(c, c->hull_in, q.slot, 0, q.B, &q.bad_mesh, &q.hull_ms, &q.err, true);
    if (q.rc == SH_OK && can_upload) {
      // the records go to the device as soon as the running step no longer reads the hull.* buffers (after its k_obb_pick)
      if (hipStreamWaitEvent(c->copy_stream, c->obb_done_ev, 0) == hipSuccess && hull_upload(c, q.slot, q.B, dst.p, c->copy_stream) == hipSuccess &&
          hipStreamSynchronize(c->copy_stream) == hipSuccess)
        q.uploaded = true;
    }
  });
}

int sh_set_overlap(sh_ctx* c, int on) {
  if (!c) return SH_ERR_ARG;
  c->overlap = on != 0;      // hulls already in preparation stay usable by the next run
  return SH_OK;
}

int sh_set_unet_turns(sh_ctx* c, int on) {
  if (!c) return SH_ERR_ARG;
  c->unet_turn = on != 0;
  if (!c->unet_turn) unet_turn_forget(c);
  return SH_OK;
}

int sh_discard_prepared(sh_ctx* c) {
  if (!c) return SH_ERR_ARG;
  (void)join_prepared(c);
  c->prep.gen = ~0ull;
  return SH_OK;
}

// ---- records on the wire ------------------------------------------------------------------------------------------------------
// A full record is 104 KB, 96 KB of it the padded anatomic-neck point list (4 096 rows; a humerus has ~1 000).  With
// sh_set_record_rows(R) the records a run hands out (`out` of sh_run / sh_submit: host memory or a gather's device send buffer)
// are PACKED: [the bytes of sh_landmarks in front of anp_points][its six trailing int32 fields][R rows of anp_points] -- the
// first min(n_anp, R) rows, zeros behind them; n_anp keeps the true count, sh_anp_points returns every row.
#define SH_REC_HEAD offsetof(sh_landmarks, anp_points)
#define SH_REC_TAIL (sizeof(sh_landmarks) - SH_REC_HEAD - sizeof(((sh_landmarks*)0)->anp_points))
static inline size_t rec_bytes_rows(int rows) { return rows > 0 ? SH_REC_HEAD + SH_REC_TAIL + (size_t)rows * 24 : sizeof(sh_landmarks); }

__global__ void __launch_bounds__(256)
k_wire_records(const sh_landmarks* __restrict__ lm, unsigned char* __restrict__ dst, int R, size_t rec) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* src = (const unsigned long long*)(lm + b);
  unsigned long long* d = (unsigned long long*)(dst + (size_t)b * rec);
  constexpr int HEAD = (int)(SH_REC_HEAD / 8), TAIL = (int)(SH_REC_TAIL / 8), PTS = (int)(sizeof(((sh_landmarks*)0)->anp_points) / 8);
  for (int i = tid; i < HEAD; i += 256) d[i] = src[i];
  for (int i = tid; i < TAIL; i += 256) d[HEAD + i] = src[HEAD + PTS + i];
  int n = lm[b].n_anp;
  n = n < 0 ? 0 : (n > R ? R : n);
  for (int i = tid; i < 3 * R; i += 256) d[HEAD + TAIL + i] = i < 3 * n ? src[HEAD + i] : 0ull;
}

// records [b0, b0 + n) of the run just enqueued -> dst (device memory, record b at dst + b * rec) on the context's stream
static int emit_records(sh_ctx* c, void* dst, int b0, int n, int rows, size_t rec) {
  const sh_landmarks* src = (const sh_landmarks*)c->bufs["landmarks"].p + b0;
  if (rows <= 0) { HIPCHK(c, hipMemcpyAsync((char*)dst + (size_t)b0 * rec, src, (size_t)n * rec, hipMemcpyDeviceToDevice, c->stream)); return SH_OK; }
  LAUNCH(c, "k_wire_records", k_wire_records, dim3(n), dim3(256), src, (unsigned char*)dst + (size_t)b0 * rec, rows, rec);
  return SH_OK;
}

// every anatomic-neck point of humerus b (CT) of the last run: the rows a packed record cut off, or all of them
__global__ void k_anp_points_ct(const double* __restrict__ pts_obb, const double* __restrict__ T_obb, int b, int n, double* __restrict__ out) {
  double Ti[16];
  inv_transform(T_obb + 16 * b, Ti);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double* p = pts_obb + ((size_t)b * SH_ANP_CAP + i) * 3;
    xform_pt(Ti, p[0], p[1], p[2], out + 3 * (size_t)i);
  }
}

// ---- the staging side of the mesh slot: a stream of NEW batches at the resident rate -------------------------------------------
// The reference's unit of work is a new STL (mesh.py:22-27, bone.py:110-131).  sh_upload_* hands a batch over synchronously:
// host-side validation, pageable copies, a stream synchronize, and hulls that can only start once the run is submitted.  The
// staging calls below do the same hand-over beside a run of the resident batch:
//   sh_stage_meshes / sh_stage_stl   copy the caller's arrays / files through page-locked staging (worker threads) or straight
//                    from page-locked caller memory, enqueue the H2D copies, the validation (on the device) -- for STL files
//                    the parse / vertex-merge kernels of k_stl.h -- and the hull prefilter on the COPY stream, start the
//                    background hull thread, and return;
//   sh_commit_staged waits for the copies (long done when a run was in flight meanwhile), swaps the buffer entries of the two
//                    sides and makes the staged batch the resident one; the next sh_submit finds its hulls prepared.
// One batch can be staged at a time; commit needs the context idle (sh_collect first): the buffers that become the staging side
// are the ones the collected run read.  A rejected batch (bad index, NaN, not an STL) is reported by sh_commit_staged and leaves
// the resident batch untouched.  Records are identical to sh_upload_* + sh_run: same device buffers, same kernels.
static void discard_staged(sh_ctx* c) {
  sh_ctx::StageSide& S = c->stg;
  if (!S.active) return;
  if (c->prep.staged) { (void)join_prepared(c); c->prep.gen = ~0ull; c->prep.staged = false; }
  (void)hipEventSynchronize(S.ready_ev);      // nothing reads the pinned staging or writes the staging side any more
  S.active = false;
}

// The caller's (pageable) memory -> page-locked staging -> device, chunk by chunk on a few threads of their own (the hull pool's
// workers may all be inside another lane's hull phase): a thread copies a chunk and enqueues its H2D copy at once, so the PCIe
// transfer runs behind the memcpy instead of after it.  37 MB of arrays / 104 MB of files per batch.
static hipError_t staged_h2d(void* dev, void* pinned, const void* src, size_t n, hipStream_t st) {
  const size_t chunk = (size_t)4 << 20;
  const size_t nch = (n + chunk - 1) / chunk;
  if (nch == 0) return hipSuccess;
  std::atomic<size_t> next(0);
  std::atomic<int> err((int)hipSuccess);
  auto work = [&]() {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= nch) break;
      const size_t o = k * chunk, m = std::min(chunk, n - o);
      memcpy((char*)pinned + o, (const char*)src + o, m);
      const hipError_t e = hipMemcpyAsync((char*)dev + o, (char*)pinned + o, m, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) err.store((int)e);
    }
  };
  static const int nthreads = [] { const char* e = getenv("SHOULDER_COPY_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads && (size_t)t < nch; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  return (hipError_t)err.load();
}

static bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at{};
  const bool pinned = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
  (void)hipGetLastError();
  return pinned;
}

static int stage_common_alloc(sh_ctx* c, int B, long long sumV_cap, long long sumF_cap, size_t flag_ints) {
  sh_ctx::StageSide& S = c->stg;
  int rc;
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!S.ready_ev) HIPCHK(c, hipEventCreateWithFlags(&S.ready_ev, hipEventDisableTiming));
  if (S.h_flag_cap < flag_ints) {
    if (S.h_flag) (void)hipHostFree(S.h_flag);
    S.h_flag = nullptr; S.h_flag_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&S.h_flag, flag_ints * 4));
    S.h_flag_cap = flag_ints;
  }
  if (S.h_kept_cap < sumV_cap) {
    if (S.h_kept) (void)hipHostFree(S.h_kept);
    S.h_kept = nullptr; S.h_kept_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&S.h_kept, (size_t)sumV_cap * 12));
    S.h_kept_cap = sumV_cap;
  }
  if (S.h_koff_cap < B + 1) {
    if (S.h_koff) (void)hipHostFree(S.h_koff);
    S.h_koff = nullptr; S.h_koff_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&S.h_koff, (size_t)(B + 1) * 8));
    S.h_koff_cap = B + 1;
  }
#define ENSS(name, bytes, elem) do { if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc; c->bufs[name].per_mesh = 0; } while (0)
  ENSS("verts.s", sumV_cap * 12, 4);
  ENSS("faces.s", sumF_cap * 12, 4);
  ENSS("voff.s", (size_t)(B + 1) * 8, 8);
  ENSS("foff.s", (size_t)(B + 1) * 8, 8);
  ENSS("stage.flag", 64, 4);
  ENSS("hullpre.ext.s", (size_t)B * SH_HP_NDIR * 4, 4);
  ENSS("hullpre.planes.s", (size_t)B * SH_HP_MAXPL * 4 * 8, 8);
  ENSS("hullpre.npl.s", (size_t)B * 4, 4);
  ENSS("hullpre.nkept.s", (size_t)B * 4, 4);
  ENSS("hullpre.pval.s", (size_t)B * SH_HP_PARTS * SH_HP_NDIR * 8, 8);
  ENSS("hullpre.pidx.s", (size_t)B * SH_HP_PARTS * SH_HP_NDIR * 4, 4);
  ENSS("hullpre.pcnt.s", (size_t)B * SH_HP_PARTS * 4, 4);
  ENSS("hullpre.poff.s", (size_t)B * SH_HP_PARTS * 8, 8);
  ENSS("hullpre.koff.s", (size_t)(B + 1) * 8, 8);
  ENSS("hullpre.kept.s", (size_t)sumV_cap * 12, 4);
#undef ENSS
  return SH_OK;
}

static HullPre hullpre_ptrs_staged(sh_ctx* c) {      // calling thread only (buffer map)
  return HullPre{(const float*)c->bufs["verts.s"].p, (const long long*)c->bufs["voff.s"].p, (int*)c->bufs["hullpre.ext.s"].p, (double*)c->bufs["hullpre.planes.s"].p,
                 (int*)c->bufs["hullpre.npl.s"].p, (float*)c->bufs["hullpre.kept.s"].p, (int*)c->bufs["hullpre.nkept.s"].p, (long long*)c->bufs["hullpre.koff.s"].p,
                 (double*)c->bufs["hullpre.pval.s"].p, (int*)c->bufs["hullpre.pidx.s"].p, (int*)c->bufs["hullpre.pcnt.s"].p, (long long*)c->bufs["hullpre.poff.s"].p};
}

// what the STL variant's thread does first (phase 1): sizes of the merged meshes -> offsets -> k_stl_emit
struct StlPhase { bool on = false; int tsize = 0; long long maxc = 0; void *corners, *coff, *table, *slot, *vid, *fpos, *voff_d, *foff_d, *verts_d, *faces_d; hipEvent_t counted = nullptr; };

// The background thread of a staged batch: (STL: phase 1,) hull points through the prefilter, host hulls into pinned slot
// `prep.slot`, records to the device as soon as the run in flight no longer reads hull.*.  `hulls` false (device hull): phase 1 only.
static void start_prepare_staged(sh_ctx* c, bool hulls, const StlPhase stl, std::function<int(std::string*)> phase0) {
  sh_ctx::Prepared& p = c->prep;
  sh_ctx::StageSide& S = c->stg;
  p.active = true; p.staged = true; p.slot = c->hslot; p.B = S.B; p.gen = hulls ? c->batch_gen + 1 : ~0ull; p.rc = SH_OK; p.bad_mesh = -1; p.d2h_ms = p.hull_ms = 0; p.err.clear();
  p.uploaded = false;
  const HullPre hp = hullpre_ptrs_staged(c);
  struct Dst { void* p[6]; } dst = {{c->bufs["hull.hv"].p, c->bufs["hull.normals"].p, c->bufs["hull.edges"].p, c->bufs["hull.nv"].p, c->bufs["hull.nf"].p, c->bufs["hull.ne"].p}};
  // early upload only into buffers that will not be re-allocated by the commit (alloc_batch grows them for a larger batch)
  const size_t nB = (size_t)S.B;
  const bool can_upload = c->obb_done_ev != nullptr && dst.p[0] && c->bufs["hull.hv"].bytes >= nB * c->hcap.v * 24 && c->bufs["hull.normals"].bytes >= nB * c->hcap.f * 24 &&
                          c->bufs["hull.edges"].bytes >= nB * c->hcap.e * 16 && c->bufs["hull.nv"].bytes >= nB * 4 && c->bufs["hull.nf"].bytes >= nB * 4 && c->bufs["hull.ne"].bytes >= nB * 4;
  const int B = S.B;
  p.th = std::thread([c, hp, dst, can_upload, hulls, stl, B, phase0]() {
    sh_ctx::Prepared& q = c->prep;
    sh_ctx::StageSide& S = c->stg;
    auto meta = [&](int rc, const char* msg) {
      { std::lock_guard<std::mutex> lk(S.m); S.meta_rc = rc; if (msg) S.meta_err = msg; S.meta_ready = true; }
      S.cv.notify_all();
    };
    if (hipSetDevice(c->device) != hipSuccess) { q.rc = SH_ERR_HIP; meta(SH_ERR_HIP, "hipSetDevice"); return; }
    const bool dbg = c->sw.debug;
    const auto tt0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(); };
    double t_p0 = 0, t_meta = 0, t_pts = 0;
    {      // phase 0: the caller's memory -> page-locked staging, the copies and the first kernels enqueued (the staging call has returned)
      std::string et;
      const int rc0 = phase0(&et);
      if (rc0 != SH_OK) { q.rc = rc0; meta(rc0, et.c_str()); return; }
      if (!stl.on) meta(SH_OK, nullptr);
      t_p0 = since();
    }
    long long sumV = S.sumV;
    if (stl.on) {
      if (hipEventSynchronize(stl.counted) != hipSuccess) { q.rc = SH_ERR_HIP; meta(SH_ERR_HIP, "sh_stage_stl: the parse kernels failed"); return; }
      const int* counts = S.h_flag + 16;            // [2 B] merged vertices, faces per file
      const int* nonfin = S.h_flag + 16 + 2 * B;    // [B]
      S.voff.assign(B + 1, 0); S.foff.assign(B + 1, 0);
      long long maxV = 0, maxF = 0;
      for (int b = 0; b < B; ++b) {
        if (nonfin[b]) { q.rc = SH_ERR_ARG; meta(SH_ERR_ARG, "sh_stage_stl: a file holds NaN / infinite coordinates"); return; }
        if (counts[2 * b] < 4 || counts[2 * b + 1] < 4) { q.rc = SH_ERR_ARG; meta(SH_ERR_ARG, "sh_stage_stl: a mesh has fewer than 4 vertices/faces after merging"); return; }
        S.voff[b + 1] = S.voff[b] + counts[2 * b];
        S.foff[b + 1] = S.foff[b] + counts[2 * b + 1];
        maxV = std::max<long long>(maxV, counts[2 * b]); maxF = std::max<long long>(maxF, counts[2 * b + 1]);
      }
      S.maxV = maxV; S.maxF = maxF; S.sumV = S.voff[B]; S.sumF = S.foff[B];
      sumV = S.sumV;
      hipStream_t st = c->copy_stream;
      bool ok = hipMemcpyAsync(stl.voff_d, S.voff.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
                hipMemcpyAsync(stl.foff_d, S.foff.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, st) == hipSuccess;
      if (ok) {
        const dim3 gc((unsigned)std::min<long long>((stl.maxc + 255) / 256, 1024), (unsigned)B);
        hipLaunchKernelGGL(k_stl_emit, gc, dim3(256), 0, st, (const float*)stl.corners, (const long long*)stl.coff, (const int2*)stl.table, stl.tsize, (const int*)stl.slot,
                           (const int*)stl.vid, (const int*)stl.fpos, (const long long*)stl.voff_d, (const long long*)stl.foff_d, (float*)stl.verts_d, (int*)stl.faces_d);
        ok = hipGetLastError() == hipSuccess && hipEventRecord(S.ready_ev, st) == hipSuccess;
      }
      if (!ok) { q.rc = SH_ERR_HIP; meta(SH_ERR_HIP, "sh_stage_stl: enqueueing the merge failed"); return; }
      meta(SH_OK, nullptr);
      t_meta = since();
    }
    if (!hulls) return;
    {
      auto t0 = std::chrono::steady_clock::now();
      if (fetch_prefiltered(hp, B, sumV, S.h_koff, S.h_kept, &S.pts, c->copy_stream) != hipSuccess) { q.rc = SH_ERR_HIP; return; }
      q.d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    t_pts = since();
    if (!stl.on && S.h_flag[0] != 0) { q.rc = SH_ERR_ARG; return; }      // (the validation word came back in front of the survivors; the commit reports it)
    q.rc = hull_host_phase(c, S.pts, q.slot, 0, B, &q.bad_mesh, &q.hull_ms, &q.err, true);
    if (dbg) fprintf(stderr, "[sh] staged batch: copies enqueued %.2f ms, sizes known %.2f, hull points back %.2f, hulls done %.2f (hull phase %.2f)\n", t_p0, t_meta, t_pts, since(), q.hull_ms);
    if (q.rc == SH_OK && can_upload) {
      if (hipStreamWaitEvent(c->copy_stream, c->obb_done_ev, 0) == hipSuccess && hull_upload(c, q.slot, B, dst.p, c->copy_stream) == hipSuccess &&
          hipStreamSynchronize(c->copy_stream) == hipSuccess)
        q.uploaded = true;
    }
  });
}

int sh_stage_meshes(sh_ctx* c, const float* verts, const int32_t* faces, const int64_t* v_off, const int64_t* f_off, int B) {
  if (!c || !verts || !faces || !v_off || !f_off || B <= 0) return fail(c, SH_ERR_ARG, "sh_stage_meshes: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  if (v_off[0] != 0 || f_off[0] != 0) return fail(c, SH_ERR_ARG, "sh_stage_meshes: offsets must start at 0");
  long long maxV = 0, maxF = 0;
  for (int b = 0; b < B; ++b) {
    const long long nv = v_off[b + 1] - v_off[b], nf = f_off[b + 1] - f_off[b];
    if (nv < 4 || nf < 4) return fail(c, SH_ERR_ARG, "sh_stage_meshes: a mesh has fewer than 4 vertices/faces");
    if (nv > 0x7fffffffLL / 3 || nf > 0x7fffffffLL / 3) return fail(c, SH_ERR_ARG, "sh_stage_meshes: a mesh is too large");
    maxV = std::max(maxV, nv); maxF = std::max(maxF, nf);
  }
  discard_staged(c);
  (void)join_prepared(c);      // one background job per context
  sh_ctx::StageSide& S = c->stg;
  const long long sumV = v_off[B], sumF = f_off[B];
  int rc;
  if ((rc = stage_common_alloc(c, B, sumV, sumF, 64)) != SH_OK) return rc;
  // the caller's arrays -> page-locked memory (unless they are page-locked already) -> device, all by the background thread: the
  // caller keeps its arrays unchanged until sh_commit_staged has returned
  const size_t vb = (size_t)sumV * 12, fb = (size_t)sumF * 12, vpad = (vb + 255) & ~(size_t)255;
  const bool vpin = is_pinned_host(verts), fpin = is_pinned_host(faces);
  const size_t need = (vpin ? 0 : vpad) + (fpin ? 0 : fb);
  if (S.h_src_cap < need) {
    if (S.h_src) (void)hipHostFree(S.h_src);
    S.h_src = nullptr; S.h_src_cap = 0;
    HIPCHK(c, hipHostMalloc(&S.h_src, need + need / 8));
    S.h_src_cap = need + need / 8;
  }
  S.B = B; S.sumV = sumV; S.sumF = sumF; S.maxV = maxV; S.maxF = maxF; S.from_stl = false;
  S.voff.assign(v_off, v_off + B + 1); S.foff.assign(f_off, f_off + B + 1);
  S.meta_ready = false; S.meta_rc = SH_OK; S.meta_err.clear();
  S.h_flag[0] = 0;
  struct P0 { sh_ctx* c; const float* verts; const int32_t* faces; size_t vb, fb, vpad; bool vpin, fpin; void *dv, *df, *dvo, *dfo; int* flag; int B; long long maxV, maxF; } a =
      {c, verts, faces, vb, fb, vpad, vpin, fpin, c->bufs["verts.s"].p, c->bufs["faces.s"].p, c->bufs["voff.s"].p, c->bufs["foff.s"].p, (int*)c->bufs["stage.flag"].p, B, maxV, maxF};
  auto phase0 = [a](std::string* et) -> int {
    sh_ctx::StageSide& S = a.c->stg;
    hipStream_t st = a.c->copy_stream;
#define P0CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *et = std::string(#call) + ": " + hipGetErrorString(e_); return SH_ERR_HIP; } } while (0)
    if (a.vpin) P0CHK(hipMemcpyAsync(a.dv, a.verts, a.vb, hipMemcpyHostToDevice, st));
    else P0CHK(staged_h2d(a.dv, S.h_src, a.verts, a.vb, st));
    if (a.fpin) P0CHK(hipMemcpyAsync(a.df, a.faces, a.fb, hipMemcpyHostToDevice, st));
    else P0CHK(staged_h2d(a.df, (char*)S.h_src + (a.vpin ? 0 : a.vpad), a.faces, a.fb, st));
    P0CHK(hipMemcpyAsync(a.dvo, S.voff.data(), (size_t)(a.B + 1) * 8, hipMemcpyHostToDevice, st));
    P0CHK(hipMemcpyAsync(a.dfo, S.foff.data(), (size_t)(a.B + 1) * 8, hipMemcpyHostToDevice, st));
    P0CHK(hipMemsetAsync(a.flag, 0, 4, st));
    hipLaunchKernelGGL(k_validate_meshes, dim3((unsigned)std::min<long long>((3 * std::max(a.maxV, a.maxF) + 255) / 256, 256), (unsigned)a.B), dim3(256), 0, st,
                       (const float*)a.dv, (const int*)a.df, (const long long*)a.dvo, (const long long*)a.dfo, a.flag);
    P0CHK(hipGetLastError());
    P0CHK(hipMemcpyAsync(S.h_flag, a.flag, 4, hipMemcpyDeviceToHost, st));
    P0CHK(hipEventRecord(S.ready_ev, st));
#undef P0CHK
    return SH_OK;
  };
  HIPCHK(c, hipEventRecord(S.ready_ev, c->copy_stream));      // (a discard before the thread gets there waits for this one)
  S.active = true;
  start_prepare_staged(c, !device_hull_now(c), StlPhase{}, phase0);
  return SH_OK;
}

int sh_stage_stl(sh_ctx* c, const void* const* files, const size_t* nbytes, int B) {
  if (!c || !files || !nbytes || B <= 0) return fail(c, SH_ERR_ARG, "sh_stage_stl: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<long long> file_off(B + 1, 0), coff(B + 1, 0);
  long long maxc = 0;
  for (int b = 0; b < B; ++b) {
    if (!files[b] || nbytes[b] < 84) return fail(c, SH_ERR_ARG, "sh_stage_stl: a file is too short for a binary STL");
    uint32_t nt;
    memcpy(&nt, (const char*)files[b] + 80, 4);
    if (nbytes[b] != 84 + 50ull * nt) return fail(c, SH_ERR_ARG, "sh_stage_stl: not a binary STL (size does not match the triangle count)");
    if (nt < 4 || nt > 0x7fffffffu / 3) return fail(c, SH_ERR_ARG, "sh_stage_stl: a mesh has fewer than 4 (or too many) triangles");
    file_off[b + 1] = file_off[b] + (long long)((nbytes[b] + 3) & ~(size_t)3);
    coff[b + 1] = coff[b] + 3ll * nt;
    maxc = std::max(maxc, 3ll * nt);
  }
  discard_staged(c);
  (void)join_prepared(c);
  sh_ctx::StageSide& S = c->stg;
  int tsize = 1024;
  while (tsize < 2 * maxc) tsize <<= 1;
  const long long sumC = coff[B];
  int rc;
  // the merged meshes are at most as large as the corner lists: the staging side is sized by that bound, the true offsets are
  // made by the thread once the device has counted
  if ((rc = stage_common_alloc(c, B, sumC, sumC / 3, 16 + 3 * (size_t)B)) != SH_OK) return rc;
#define ENSS(name, bytes, elem) do { if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc; c->bufs[name].per_mesh = 0; } while (0)
  ENSS("stl.raw", (size_t)file_off[B], 1);
  ENSS("stl.file_off", (B + 1) * 8, 8);
  ENSS("stl.coff", (B + 1) * 8, 8);
  ENSS("stl.corners", (size_t)sumC * 12, 4);
  ENSS("stl.table", (size_t)B * tsize * 8, 4);
  ENSS("stl.slot", (size_t)sumC * 4, 4);
  ENSS("stl.vid", (size_t)sumC * 4, 4);
  ENSS("stl.fpos", (size_t)(sumC / 3) * 4, 4);
  ENSS("stl.counts", (size_t)B * 8, 4);
  ENSS("stl.bsum", stl_rank_scratch_ints(B, maxc) * 4, 4);
  ENSS("stl.nonfinite", (size_t)B * 4, 4);
#undef ENSS
  // files -> one page-locked image (file starts 4-byte aligned) + the two offset tables behind it
  const size_t raw_bytes = (size_t)file_off[B], tab_off = (raw_bytes + 255) & ~(size_t)255, need = tab_off + 2 * (size_t)(B + 1) * 8;
  if (S.h_src_cap < need) {
    if (S.h_src) (void)hipHostFree(S.h_src);
    S.h_src = nullptr; S.h_src_cap = 0;
    HIPCHK(c, hipHostMalloc(&S.h_src, need + need / 8));
    S.h_src_cap = need + need / 8;
  }
  long long* h_tabs = (long long*)((char*)S.h_src + tab_off);
  memcpy(h_tabs, file_off.data(), (size_t)(B + 1) * 8);
  memcpy(h_tabs + B + 1, coff.data(), (size_t)(B + 1) * 8);
  S.B = B; S.from_stl = true; S.sumV = S.sumF = S.maxV = S.maxF = 0;
  S.meta_ready = false; S.meta_rc = SH_OK; S.meta_err.clear();
  S.h_flag[0] = 0;
  if (!c->stl_counted_ev) HIPCHK(c, hipEventCreateWithFlags(&c->stl_counted_ev, hipEventDisableTiming));
  // the files themselves are read by the background thread: the caller keeps them unchanged until sh_commit_staged has returned
  struct P0 { sh_ctx* c; std::vector<const void*> files; std::vector<size_t> nbytes; std::vector<long long> file_off; std::vector<char> pinned; size_t raw_bytes; long long* h_tabs; int B, tsize; long long maxc;
              unsigned char* raw; void *d_file_off, *d_coff, *corners, *table, *slot, *vid, *fpos, *counts, *nonfin, *bsum; };
  auto a = std::make_shared<P0>();
  a->pinned.resize(B);
  for (int b = 0; b < B; ++b) a->pinned[b] = is_pinned_host(files[b]) ? 1 : 0;
  a->c = c; a->files.assign(files, files + B); a->nbytes.assign(nbytes, nbytes + B); a->file_off = file_off; a->raw_bytes = raw_bytes; a->h_tabs = h_tabs; a->B = B; a->tsize = tsize; a->maxc = maxc;
  a->raw = (unsigned char*)c->bufs["stl.raw"].p; a->d_file_off = c->bufs["stl.file_off"].p; a->d_coff = c->bufs["stl.coff"].p; a->corners = c->bufs["stl.corners"].p;
  a->table = c->bufs["stl.table"].p; a->slot = c->bufs["stl.slot"].p; a->vid = c->bufs["stl.vid"].p; a->fpos = c->bufs["stl.fpos"].p; a->counts = c->bufs["stl.counts"].p;
  a->nonfin = c->bufs["stl.nonfinite"].p; a->bsum = c->bufs["stl.bsum"].p;
  auto phase0 = [a](std::string* et) -> int {
    sh_ctx* c = a->c;
    sh_ctx::StageSide& S = c->stg;
    hipStream_t st = c->copy_stream;
    const int B = a->B;
#define P0CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *et = std::string(#call) + ": " + hipGetErrorString(e_); return SH_ERR_HIP; } } while (0)
    {      // file by file: page-locked files go as they are, the others through the staging image; each file's H2D follows its memcpy at once
      std::atomic<int> next(0);
      std::atomic<int> err((int)hipSuccess);
      auto work = [&]() {
        for (;;) {
          const int b = next.fetch_add(1);
          if (b >= B) break;
          const void* src = a->files[b];
          if (!a->pinned[b]) { memcpy((char*)S.h_src + a->file_off[b], a->files[b], a->nbytes[b]); src = (char*)S.h_src + a->file_off[b]; }
          const hipError_t e = hipMemcpyAsync(a->raw + a->file_off[b], src, a->nbytes[b], hipMemcpyHostToDevice, st);
          if (e != hipSuccess) err.store((int)e);
        }
      };
      static const int nthreads = [] { const char* e = getenv("SHOULDER_COPY_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
      std::vector<std::thread> th;
      for (int t = 1; t < nthreads && t < B; ++t) th.emplace_back(work);
      work();
      for (auto& t : th) t.join();
      P0CHK((hipError_t)err.load());
    }
    P0CHK(hipMemcpyAsync(a->d_file_off, a->h_tabs, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, st));
    P0CHK(hipMemcpyAsync(a->d_coff, a->h_tabs + B + 1, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, st));
    P0CHK(hipMemsetAsync(a->nonfin, 0, (size_t)B * 4, st));
    const dim3 gc((unsigned)std::min<long long>((a->maxc + 255) / 256, 1024), (unsigned)B);
    hipLaunchKernelGGL(k_stl_corners, gc, dim3(256), 0, st, (const unsigned char*)a->raw, (const long long*)a->d_file_off, (const long long*)a->d_coff, (float*)a->corners, (int*)a->nonfin);
    hipLaunchKernelGGL(k_stl_table_init, dim3(1024), dim3(256), 0, st, (int2*)a->table, (size_t)B * a->tsize);
    hipLaunchKernelGGL(k_stl_hash, gc, dim3(256), 0, st, (const float*)a->corners, (const long long*)a->d_coff, (int2*)a->table, a->tsize, (int*)a->slot);
    stl_rank_launch(st, B, a->maxc, (const long long*)a->d_coff, (const int2*)a->table, a->tsize, (const int*)a->slot, (int*)a->vid, (int*)a->fpos, (int*)a->counts, (int*)a->bsum);
    P0CHK(hipGetLastError());
    P0CHK(hipMemcpyAsync(S.h_flag + 16, a->counts, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    P0CHK(hipMemcpyAsync(S.h_flag + 16 + 2 * B, a->nonfin, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    P0CHK(hipEventRecord(c->stl_counted_ev, st));
#undef P0CHK
    return SH_OK;
  };
  StlPhase ph;
  ph.on = true; ph.tsize = tsize; ph.maxc = maxc;
  ph.corners = c->bufs["stl.corners"].p; ph.coff = c->bufs["stl.coff"].p; ph.table = c->bufs["stl.table"].p; ph.slot = c->bufs["stl.slot"].p;
  ph.vid = c->bufs["stl.vid"].p; ph.fpos = c->bufs["stl.fpos"].p; ph.voff_d = c->bufs["voff.s"].p; ph.foff_d = c->bufs["foff.s"].p;
  ph.verts_d = c->bufs["verts.s"].p; ph.faces_d = c->bufs["faces.s"].p;
  HIPCHK(c, hipEventRecord(S.ready_ev, c->copy_stream));      // (recorded again behind k_stl_emit by the thread; this one covers an early discard)
  ph.counted = c->stl_counted_ev;
  S.active = true;
  start_prepare_staged(c, !device_hull_now(c), ph, phase0);
  return SH_OK;
}

int sh_commit_staged(sh_ctx* c, int64_t* v_off_out, int64_t* f_off_out) {
  if (!c) return SH_ERR_ARG;
  sh_ctx::StageSide& S = c->stg;
  if (!S.active) return fail(c, SH_ERR_STATE, "sh_commit_staged: no batch is staged");
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_commit_staged: runs are in flight (sh_collect them first: they read the buffers that become the staging side)");
  HIPCHK(c, hipSetDevice(c->device));
  { std::unique_lock<std::mutex> lk(S.m); S.cv.wait(lk, [&] { return S.meta_ready; }); }
  if (S.meta_rc != SH_OK) { const int rc = S.meta_rc; const std::string msg = S.meta_err; discard_staged(c); return fail(c, rc, msg); }
  HIPCHK(c, hipEventSynchronize(S.ready_ev));
  if (!S.from_stl && S.h_flag[0] != 0) {
    const int f = S.h_flag[0];
    discard_staged(c);
    return fail(c, SH_ERR_ARG, (f & 1) ? "sh_stage_meshes: face index out of range" : "sh_stage_meshes: NaN / infinite vertex coordinate");
  }
  // a resident-overlap preparation cannot be under way (staging joined it); the staged batch's own thread keeps running
  for (const char* nm : {"verts", "faces", "voff", "foff"}) std::swap(c->bufs[nm], c->bufs[std::string(nm) + ".s"]);
  c->B = 0;
  c->h_voff.swap(S.voff); c->h_foff.swap(S.foff);
  c->sumV = S.sumV; c->sumF = S.sumF; c->maxV = S.maxV; c->maxF = S.maxF;
  c->h_verts_valid = false;
  ++c->batch_gen;
  c->bufs["verts"].per_mesh = 0; c->bufs["faces"].per_mesh = 0;
  c->bufs["verts.s"].per_mesh = 0; c->bufs["faces.s"].per_mesh = 0; c->bufs["voff.s"].per_mesh = 0; c->bufs["foff.s"].per_mesh = 0;
  const int B = S.B;
  S.active = false;
  if (v_off_out) for (int b = 0; b <= B; ++b) v_off_out[b] = c->h_voff[b];
  if (f_off_out) for (int b = 0; b <= B; ++b) f_off_out[b] = c->h_foff[b];
  c->B = B;
  int rc = alloc_batch(c);
  if (rc != SH_OK) c->B = 0;
  return rc;
}

int sh_staged(const sh_ctx* c) { return c ? (c->stg.active ? 1 : 0) : SH_ERR_ARG; }

// The device hull gave the humeri in `list` up during the run of ticket `tk` (k_hull.h writes a unit tetrahedron for them, so
// everything queued behind ran on finite data and their records are void).  Each of them gets the host quickhull -- which
// has the retry / joggle logic -- its record goes into the device buffers where the device hull would have put it, and its
// stages run again as a window of one humerus.  All of it is enqueued on the context's stream: behind a second run that may
// be in flight (which finished with this batch's scratch buffers by then, and has parked its own results per ticket).
// hulld.skip[b] is set, so later runs of the resident batch get these humeri right the first time.
static int redo_given_up(sh_ctx* c, sh_ctx::Ticket& tk, const std::string& tslot, const std::vector<int>& list, unsigned long long* need /*[4]: pool demand, [3] = rerun*/) {
  const int B = tk.B;
  std::vector<float> hv32;
  std::vector<double> P;
  shhull::Hull H;
  int rc = SH_OK;
  for (int b : list) {
    const long long v0 = c->h_voff[b], nv = c->h_voff[b + 1] - v0;
    hv32.resize(3 * (size_t)nv);
    HIPCHK(c, hipMemcpyAsync(hv32.data(), (const float*)c->bufs["verts"].p + 3 * v0, (size_t)nv * 12, hipMemcpyDeviceToHost, c->out_stream));
    HIPCHK(c, hipStreamSynchronize(c->out_stream));
    P.assign(hv32.begin(), hv32.end());
    int status = 0, hn = 0, fn = 0, en = 0;
    if (!shhull::convex_hull(P.data(), (int)nv, H)) status = SH_ERR_GEOMETRY;
    else {
      hn = (int)H.vert_ids.size(); fn = (int)H.tris.size() / 3; en = (int)H.edges.size() / 4;
      if (hn > c->hcap.v || fn > c->hcap.f || en > c->hcap.e) {
        // above the record: growing it here would drop the other humeri's records -- the batch runs again with its hulls from the
        // host (hull_host_phase sizes the staging, run_obb grows the record), and stays there while it is resident
        c->hull_force_host = true;
        if (need) need[3] = 1;
        return SH_OK;
      }
    }
    if (status != 0) { char m[96]; snprintf(m, sizeof m, "mesh %d: convex hull failed (%d)", b, status); return fail(c, status, m); }
    std::vector<double> hvd(3 * (size_t)hn);
    for (int i = 0; i < hn; ++i)
      for (int k = 0; k < 3; ++k) hvd[3 * (size_t)i + k] = P[3 * (size_t)H.vert_ids[i] + k];
    c->b0 = 0; c->Bwin = B;
    const int counts[3] = {hn, fn, en}, one = 1;
    // (pageable sources: hipMemcpyAsync stages them before it returns)
    HIPCHK(c, hipMemcpyAsync(buf<double>(c, "hull.hv") + (size_t)b * c->hcap.v * 3, hvd.data(), hvd.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<double>(c, "hull.normals") + (size_t)b * c->hcap.f * 3, H.normals.data(), (size_t)fn * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "hull.edges") + (size_t)b * c->hcap.e * 4, H.edges.data(), (size_t)en * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "hull.nv") + b, &counts[0], 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "hull.nf") + b, &counts[1], 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "hull.ne") + b, &counts[2], 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf<int>(c, "hulld.skip") + b, &one, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(buf<int>(c, "err") + b, 0, 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // the sources above are locals
    c->skip_nfmax = std::max(c->skip_nfmax, fn);
    c->b0 = b; c->Bwin = 1; c->redo_records = true; c->redo_nf = fn;
    rc = run_window(c, tk.mask, -1);
    c->redo_records = false;
    c->b0 = 0; c->Bwin = B;
    if (rc != SH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    // the record and the status word of this humerus -> where the run's results were parked (or the caller's device buffer)
    const sh_landmarks* src = buf<sh_landmarks>(c, "landmarks") + b;
    (void)src;
    if (tk.host_out) { int erc = emit_records(c, c->bufs["out.landmarks" + tslot].p, b, 1, tk.rows, tk.rec); if (erc != SH_OK) return erc; }
    else if (tk.out_arg) { int erc = emit_records(c, tk.out_arg, b, 1, tk.rows, tk.rec); if (erc != SH_OK) return erc; }
    HIPCHK(c, hipMemcpyAsync((int*)c->bufs["out.err" + tslot].p + b, buf<int>(c, "err") + b, 4, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  {
    // The one-humerus windows above took their overflow ranges from the pools the void first pass had already drawn on, past
    // sh_collect's grow-and-rerun check.  A dense humerus (the kind the device hull gives up) may have asked for more than the
    // pools hold: k_ovf_plan flagged it SH_ERR_CAPACITY_DEV.  Report the demand; sh_collect grows the pools and runs the whole
    // batch again (hulld.skip keeps the device hull off these humeri by then).
    unsigned long long ctr[8];
    HIPCHK(c, hipMemcpyAsync(ctr, c->bufs["ovf.ctr"].p, 64, hipMemcpyDeviceToHost, c->out_stream));
    HIPCHK(c, hipStreamSynchronize(c->out_stream));
    if (ctr[3] > c->ovf_seg_cap || ctr[4] > c->ovf_ring_cap || ctr[5] > c->ovf_work_cap || ctr[6] != 0) {
      if (need) { need[0] = ctr[3]; need[1] = ctr[4]; need[2] = ctr[5]; need[3] = 1; }
      return SH_OK;
    }
  }
  if (tk.host_out)
    HIPCHK(c, hipMemcpyAsync(tk.host_out, c->bufs["out.landmarks" + tslot].p, (size_t)B * tk.rec, hipMemcpyDeviceToHost, c->out_stream));
  HIPCHK(c, hipMemcpyAsync(tk.h_err, c->bufs["out.err" + tslot].p, (size_t)B * 4, hipMemcpyDeviceToHost, c->out_stream));
  HIPCHK(c, hipStreamSynchronize(c->out_stream));
  return SH_OK;
}

static inline size_t status_ovf_off(int B) { return ((size_t)B * 4 + 7) & ~(size_t)7; }
#define SH_NCTR 16
static inline size_t status_bytes(int B) { return status_ovf_off(B) + SH_NCTR * 8 + (size_t)B * 4; }
// the status block of a run (layout: sh_submit) from the live words, one launch
__global__ void k_stage_status(const int* __restrict__ err, const unsigned long long* __restrict__ ovf_ctr, const int* __restrict__ hull_fail /*or null*/,
                               char* __restrict__ dst, int B, size_t ovf_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { ((int*)dst)[i] = err[i]; ((int*)(dst + ovf_off + SH_NCTR * 8))[i] = hull_fail ? hull_fail[i] : 0; }
  if (i < SH_NCTR) ((unsigned long long*)(dst + ovf_off))[i] = ovf_ctr[i];
}

int sh_submit(sh_ctx* c, uint32_t mask, sh_landmarks* out) {
  if (!c) return SH_ERR_ARG;
  if (c->B < 1) return fail(c, SH_ERR_STATE, "sh_run: no meshes uploaded");
  if (c->n_pending >= 2) return fail(c, SH_ERR_STATE, "sh_submit: two runs are in flight already (sh_collect first)");
  HIPCHK(c, hipSetDevice(c->device));
  const int B = c->B;
  sh_ctx::Ticket& tk = c->tickets[c->t_head];
  if (tk.cap < B) {
    // the status words of a run as ONE pinned block = one device-to-host copy and one wait in sh_collect (they were three copies and
    // three waits: each a blit kernel that queues for a CU beside the other lane's UNet): [err: B ints | pad to 8 | overflow pool
    // counters: 8 x u64 | device hull's give-up words: B ints]
    if (tk.h_err) (void)hipHostFree(tk.h_err);
    tk.h_err = nullptr; tk.h_fail = nullptr; tk.h_ovf = nullptr; tk.cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&tk.h_err, status_bytes(B)));
    tk.cap = B;
  }
  tk.h_ovf = (unsigned long long*)((char*)tk.h_err + status_ovf_off(B));      // (the layout follows THIS run's batch size)
  tk.h_fail = (int*)((char*)tk.h_err + status_ovf_off(B) + SH_NCTR * 8);
  if (!tk.ev) HIPCHK(c, hipEventCreateWithFlags(&tk.ev, hipEventDisableTiming));
  {      // the other ticket's pinned block, event and device staging with the first one (else the context's SECOND run pays for them)
    sh_ctx::Ticket& to = c->tickets[c->t_head ^ 1];
    if (!to.pending && to.cap < B) {
      if (to.h_err) (void)hipHostFree(to.h_err);
      to.h_err = nullptr; to.h_fail = nullptr; to.h_ovf = nullptr; to.cap = 0;
      HIPCHK(c, hipHostMalloc((void**)&to.h_err, status_bytes(B)));
      to.cap = B;
    }
    if (!to.ev) HIPCHK(c, hipEventCreateWithFlags(&to.ev, hipEventDisableTiming));
    const std::string oslot = std::to_string(c->t_head ^ 1);
    if (!to.pending) {
      if (int e = ensure(c, ("out.err" + oslot).c_str(), status_bytes(B), 4)) return e;
      if (out) { hipPointerAttribute_t at{}; const bool dev = hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeDevice; (void)hipGetLastError();
                 if (!dev) { if (int e = ensure(c, ("out.landmarks" + oslot).c_str(), (size_t)B * rec_bytes_rows(c->rec_rows), 1)) return e; } }
    }
  }
  if (!c->out_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
  c->b0 = 0; c->Bwin = B;
  if (c->params.bone_kind == SH_BONE_PROXIMAL) {
    if (mask & (SH_STAGE_DISTAL | SH_STAGE_TE)) return fail(c, SH_ERR_ARG, "sh_run: a proximal humerus has no distal / trans-epicondylar stage (bone.py:24-64)");
    int prc = alloc_prox(c);
    if (prc != SH_OK) return prc;
  }
  { OvfPools OP; int orc = ovf_pools(c, &OP); if (orc != SH_OK) return orc; FILL(c, {buf<int>(c, "err"), (size_t)B * 4, 0}, {OP.ctr, 128, 0}); }      // overflow pools: empty, no demand recorded
  if ((mask & SH_STAGE_APPLY) && !(mask & SH_STAGE_CSYS)) return fail(c, SH_ERR_ARG, "sh_run: SH_STAGE_APPLY needs SH_STAGE_CSYS in the same run");
  // a proximal humerus' frame is canal / articular (bone.py:53-62): k_pack builds it from the anatomic-neck axes of THIS run
  if (c->params.bone_kind == SH_BONE_PROXIMAL && (mask & SH_STAGE_CSYS) && !(mask & SH_STAGE_ANP))
    return fail(c, SH_ERR_ARG, "sh_run: SH_STAGE_CSYS of a proximal humerus needs SH_STAGE_ANP in the same run (canal / articular frame)");
  if (!(mask & SH_STAGE_OBB) && !c->obb_injected)
    return fail(c, SH_ERR_STATE, "sh_run: no OBB transform (run SH_STAGE_OBB or sh_store(\"obb_transform\"))");
  // Windows: with the host hull in play the batch can be walked in windows of SHOULDER_WINDOW humeri; all device work of
  // a window is only enqueued, so the hulls of the next window are computed while it runs (off by default).
  int wsize = SH_WINDOW;
  if (c->sw.window > 0) wsize = c->sw.window;     // tests exercise small windows
  const bool dev_hull = (mask & SH_STAGE_OBB) && device_hull_now(c);
  const int win = ((mask & SH_STAGE_OBB) && B > wsize && !dev_hull) ? wsize : B;      // (windows exist to overlap HOST hulls with device work)
  // hulls prepared by the background thread during the previous run (sh_set_overlap)?
  int prepared = -1;
  if (c->prep.active) {
    int prc = join_prepared(c);
    if ((mask & SH_STAGE_OBB) && win == B && prc == SH_OK && c->prep.gen == c->batch_gen && c->prep.B == B) { prepared = c->prep.slot; c->hslot = prepared ^ 1; }
  }
  // a run of the resident batch while hulls for the STAGED one are waiting: its own hull phase takes the same pinned slot and
  // the same hull.* device buffers, so those hulls are void (the staged batch stays; its first run computes them again)
  if ((mask & SH_STAGE_OBB) && c->prep.staged && c->prep.gen != c->batch_gen) c->prep.gen = ~0ull;
  if (dev_hull) prepared = -1;
  if ((mask & SH_STAGE_OBB) && prepared < 0 && !dev_hull) {
    // the host hull needs its points (a device-generated batch: every run, a new batch is new data)
    auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, fetch_hull_points(c, hullpre_ptrs(c), c->stream));
    if (c->sw.debug) fprintf(stderr, "[sh] submit: hull points fetched after %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (c->timing && !c->h_verts_valid) { KTimer& a = c->timers["host.verts_d2h"]; a.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); a.n += 1; }
  }
  int rc = SH_OK, widx = 0;
  for (int b0 = 0; b0 < B && rc == SH_OK; b0 += win, ++widx) {
    c->b0 = b0; c->Bwin = std::min(win, B - b0);
    rc = run_window(c, mask, widx == 0 ? prepared : -1);
  }
  // everything of this run is enqueued: the host is free until the device is done -> hulls of the next run
  c->b0 = 0; c->Bwin = B;      // (in front of start_prepare: it looks the device buffers up through the window offset)
  if (rc == SH_OK && c->overlap && (mask & SH_STAGE_OBB) && win == B && !dev_hull && !c->stg.active) start_prepare(c);
  if (mask & SH_STAGE_OBB) c->obb_injected = true;
  if (rc != SH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
  // (`out` may also be device memory, e.g. the send buffer of a gather: hipMemcpyDefault)
  // Results.  A device-to-host copy enqueued HERE would sit in a DMA queue until the kernels of this run are done and hold
  // up every later copy behind it - the vertex read-back of the background hull thread above all, which then loses its
  // overlap with the device (measured: 14.3 -> 19.2 ms per step at B=64).  So the records and status words are parked in a
  // per-ticket device buffer (a 16 us device-to-device copy at the end of the run) and sh_collect copies them to the host
  // once the run has finished.  Device `out` (the send buffer of a gather) is written directly.
  hipPointerAttribute_t at{};
  const bool out_on_device = out && hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeDevice;
  (void)hipGetLastError();
  const std::string tslot = std::to_string(c->t_head);
  void *lm_stage = nullptr, *err_stage = nullptr;
  if (int e = ensure(c, ("out.err" + tslot).c_str(), status_bytes(B), 4, &err_stage)) return e;      // (the whole status block: the status words lead it)
  tk.host_out = nullptr;
  tk.rows = c->rec_rows; tk.rec = rec_bytes_rows(c->rec_rows);
  if (out_on_device) {
    if (int e = emit_records(c, out, 0, B, tk.rows, tk.rec)) return e;
  } else if (out) {
    if (int e = ensure(c, ("out.landmarks" + tslot).c_str(), (size_t)B * tk.rec, 1, &lm_stage)) return e;
    if (int e = emit_records(c, lm_stage, 0, B, tk.rows, tk.rec)) return e;
    tk.host_out = out;
  }
  // status words, what the run asked of the overflow pools (k_ovf.h: sh_collect grows them and runs again if it was more than they
  // hold) and which humeri the device hull gave up (its own word per humerus: the status word can be overwritten by a later stage)
  LAUNCH(c, "k_stage_status", k_stage_status, dim3((unsigned)((std::max(B, SH_NCTR) + 255) / 256)), dim3(256), (const int*)buf<int>(c, "err"), (const unsigned long long*)c->bufs["ovf.ctr"].p,
         dev_hull ? (const int*)buf<int>(c, "hulld.fail") : (const int*)nullptr, (char*)err_stage, B, status_ovf_off(B));
  HIPCHK(c, hipEventRecord(tk.ev, c->stream));
  tk.B = B; tk.pending = true; tk.mask = mask; tk.out_arg = out; tk.dev_hull = dev_hull; tk.gen = c->batch_gen;
  c->t_head ^= 1; ++c->n_pending;
  return SH_OK;
}

int sh_collect(sh_ctx* c) {
  if (!c) return SH_ERR_ARG;
  if (c->n_pending == 0) return fail(c, SH_ERR_STATE, "sh_collect: nothing in flight");
  HIPCHK(c, hipSetDevice(c->device));
  sh_ctx::Ticket& tk = c->tickets[c->t_tail];
  const std::string tslot = std::to_string(c->t_tail);
  c->t_tail ^= 1; --c->n_pending; tk.pending = false;
  HIPCHK(c, hipEventSynchronize(tk.ev));
  if (tk.host_out)
    HIPCHK(c, hipMemcpyAsync(tk.host_out, buf<char>(c, ("out.landmarks" + tslot).c_str()), (size_t)tk.B * tk.rec, hipMemcpyDeviceToHost, c->out_stream));
  HIPCHK(c, hipMemcpyAsync(tk.h_err, buf<char>(c, ("out.err" + tslot).c_str()), status_bytes(tk.B), hipMemcpyDeviceToHost, c->out_stream));      // status, pool counters, give-up words
  HIPCHK(c, hipStreamSynchronize(c->out_stream));
  {
    const unsigned long long need_s = tk.h_ovf[3], need_r = tk.h_ovf[4], need_w = tk.h_ovf[5];
    const uint32_t slice_stages = SH_STAGE_FULL | SH_STAGE_DISTAL | SH_STAGE_NECK | SH_STAGE_PROXIMAL;
    // (only a run that computed its own frame may vouch for the batch: a run on an injected frame says nothing about the planes of
    // the next SH_STAGE_OBB, and the `pobb` set of a proximal humerus runs inside that stage)
    if (need_r == 0 && need_s == 0 && (tk.mask & SH_STAGE_OBB) &&
        (tk.mask & slice_stages) == (c->params.bone_kind == SH_BONE_PROXIMAL ? (slice_stages & ~(uint32_t)SH_STAGE_DISTAL) : slice_stages) &&
        tk.gen == c->batch_gen)
      c->ovf_none_gen = c->batch_gen;
    if (tk.h_ovf[6] != 0) {
      // the overflow tier was skipped and a plane needed it (k_slice_link_large): the planes moved after the run that vouched for the
      // batch.  The records of this run are void -- run it again with the tier on.
      c->ovf_none_gen = ~0ull;
      if (c->n_pending != 0)
        return fail(c, SH_ERR_CAPACITY, "a section needs the overflow tier while another run is in flight: collect it, then run the batch again (the tier is on by then)");
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const uint32_t mask = tk.mask; sh_landmarks* out = tk.out_arg;
      int rc2 = sh_submit(c, mask, out);
      if (rc2 != SH_OK) return rc2;
      return sh_collect(c);
    }
    if (c->sw.debug) fprintf(stderr, "[sh] collect: ovf need %llu %llu %llu cap %llu %llu %llu err0 %d\n", need_s, need_r, need_w, c->ovf_seg_cap, c->ovf_ring_cap, c->ovf_work_cap, tk.h_err[0]);
    if ((tk.h_ovf[8] != 0 || tk.h_ovf[9] != 0) && (tk.mask & SH_STAGE_OBB) && tk.gen == c->batch_gen) {
      // k_obb_candidates met a direction with more silhouette edges than its tier's lists hold (or, behind a device hull, a record with
      // more faces than its masks): the records of this run are void -- the batch again, on the tier that holds it (run_obb)
      if (c->n_pending != 0)
        return fail(c, SH_ERR_CAPACITY, "the OBB stage needs a larger tier while another run is in flight: collect it, then run the batch again (the tier is chosen by then)");
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if ((int)tk.h_ovf[8] <= c->obb_sil_need && !(tk.h_ovf[9] != 0 && !c->obb_nf_over))
        return fail(c, SH_ERR_CAPACITY, "k_obb_candidates: silhouette demand did not shrink on the workspace tier");      // (cannot happen: its lists hold what the last run asked for)
      c->obb_sil_need = std::max(c->obb_sil_need, (int)std::min<unsigned long long>(tk.h_ovf[8], 1ull << 30));
      if (tk.h_ovf[9] != 0) c->obb_nf_over = true;
      const uint32_t mask = tk.mask; sh_landmarks* out = tk.out_arg;
      int rc2 = sh_submit(c, mask, out);
      if (rc2 != SH_OK) return rc2;
      return sh_collect(c);
    }
    const unsigned long long need_e = tk.h_ovf[7];
    if (need_s > c->ovf_seg_cap || need_r > c->ovf_ring_cap || need_w > c->ovf_work_cap || need_e > (unsigned long long)c->end_cap) {
      // The batch has more overflow planes than the pools hold (a first dense mesh): grow them to what the run asked for,
      // with headroom, and run the batch again -- here, synchronously, when nothing else is in flight.
      if (c->n_pending != 0)
        return fail(c, SH_ERR_CAPACITY, "slice overflow pools too small while another run is in flight: collect it, then run the batch again (the pools are grown by then)");
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->ovf_seg_cap = std::max(c->ovf_seg_cap, need_s + need_s / 4);
      c->ovf_ring_cap = std::max(c->ovf_ring_cap, need_r + need_r / 4);
      c->ovf_work_cap = std::max(c->ovf_work_cap, need_w + need_w / 4);
      if (need_e > (unsigned long long)c->end_cap) {      // an end section of the box with more crossing points than "obb.endpts" holds (a very dense mesh)
        if (need_e > (1ull << 26)) return fail(c, SH_ERR_CAPACITY, "an end section has more than 2^26 crossing points");
        c->end_cap = (int)(need_e + need_e / 4);
        const size_t eb = (size_t)c->B * 2 * c->end_cap * 2 * 8;
        if (int e = ensure(c, "obb.endpts", eb, 8)) return e;
        c->bufs["obb.endpts"].per_mesh = eb / (size_t)c->B;
      }
      const uint32_t mask = tk.mask; sh_landmarks* out = tk.out_arg;
      int rc2 = sh_submit(c, mask, out);
      if (rc2 != SH_OK) return rc2;
      return sh_collect(c);
    }
  }
  if (tk.dev_hull) {
    std::vector<int> gave_up;
    for (int b = 0; b < tk.B; ++b) if (tk.h_fail[b] != 0) gave_up.push_back(b);
    if (!gave_up.empty()) {
      unsigned long long need[4] = {0, 0, 0, 0};
      int rc2 = redo_given_up(c, tk, tslot, gave_up, need);
      if (rc2 != SH_OK) return rc2;
      if (need[3]) {
        if (c->n_pending != 0)
          return fail(c, SH_ERR_CAPACITY, "slice overflow pools too small while another run is in flight: collect it, then run the batch again (the pools are grown by then)");
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->ovf_none_gen = ~0ull;
        c->ovf_seg_cap = std::max(c->ovf_seg_cap, need[0] + need[0] / 4);
        c->ovf_ring_cap = std::max(c->ovf_ring_cap, need[1] + need[1] / 4);
        c->ovf_work_cap = std::max(c->ovf_work_cap, need[2] + need[2] / 4);
        const uint32_t mask = tk.mask; sh_landmarks* out = tk.out_arg;
        int rc3 = sh_submit(c, mask, out);
        if (rc3 != SH_OK) return rc3;
        return sh_collect(c);
      }
    }
  }
  for (int b = 0; b < tk.B; ++b)
    if (tk.h_err[b] != 0) {
      char m[128];
      snprintf(m, sizeof m, "mesh %d: device stage error %d (capacity=-4, geometry=-5)", b, tk.h_err[b]);
      return fail(c, tk.h_err[b], m);
    }
  return SH_OK;
}

int sh_run(sh_ctx* c, uint32_t mask, sh_landmarks* out) {
  if (!c) return SH_ERR_ARG;
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_run: submitted runs are still in flight (sh_collect them first)");
  int rc = sh_submit(c, mask, out);
  if (rc != SH_OK) return rc;
  rc = sh_collect(c);
  if (rc == SH_OK) HIPCHK(c, hipStreamSynchronize(c->stream));
  return rc;
}

// Page-locked host memory for result buffers the caller reuses from run to run (a fresh pageable buffer per run costs a
// page fault per 4 KB plus a staged copy: ~1 ms for the 6.8 MB of 64 landmark records).
int sh_host_alloc(sh_ctx* c, size_t nbytes, void** out) {
  if (!c || !out || nbytes == 0) return fail(c, SH_ERR_ARG, "sh_host_alloc: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(out, nbytes));
  return SH_OK;
}

int sh_host_free(sh_ctx* c, void* p) {
  if (!c) return SH_ERR_ARG;
  if (p) HIPCHK(c, hipHostFree(p));
  return SH_OK;
}

int sh_set_keep_products(sh_ctx* c, int on) {
  if (!c) return SH_ERR_ARG;
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_set_keep_products: runs are in flight");
  c->keep_products = on != 0;
  return SH_OK;
}

int sh_set_record_rows(sh_ctx* c, int anp_rows) {
  if (!c || anp_rows < 0 || anp_rows > SH_ANP_MAX_PTS) return fail(c, SH_ERR_ARG, "sh_set_record_rows: 0 (full records) .. 4096 rows");
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_set_record_rows: runs are in flight");
  c->rec_rows = anp_rows;
  return SH_OK;
}

size_t sh_record_bytes(int anp_rows) { return rec_bytes_rows(anp_rows); }

int sh_anp_points(sh_ctx* c, int b, double* out, int cap, int* n_out) {
  if (!c || !n_out || b < 0 || b >= c->B || cap < 0 || (cap > 0 && !out)) return fail(c, SH_ERR_ARG, "sh_anp_points: bad argument");
  if (c->n_pending != 0) return fail(c, SH_ERR_STATE, "sh_anp_points: runs are in flight (sh_collect them first)");
  if (c->bufs.find("anp.counts") == c->bufs.end()) return fail(c, SH_ERR_STATE, "sh_anp_points: no run yet");
  HIPCHK(c, hipSetDevice(c->device));
  c->b0 = 0; c->Bwin = c->B;
  int cnt[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(cnt, buf<int>(c, "anp.counts") + 2 * b, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int n = std::min(cnt[0], (int)SH_ANP_CAP);
  *n_out = n;
  n = std::min(n, cap);
  if (n <= 0) return SH_OK;
  int rc = ensure(c, "anp.points_ct_one", (size_t)SH_ANP_CAP * 24, 8);
  if (rc != SH_OK) return rc;
  c->bufs["anp.points_ct_one"].per_mesh = 0;
  double* d = (double*)c->bufs["anp.points_ct_one"].p;
  LAUNCH(c, "k_anp_points_ct", k_anp_points_ct, dim3((unsigned)((n + 255) / 256)), dim3(256), (const double*)buf<double>(c, "anp.points_obb"), (const double*)buf<double>(c, "obb_transform"), b, n, d);
  HIPCHK(c, hipMemcpyAsync(out, d, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

int sh_landmarks_device(sh_ctx* c, void** p, size_t* n) {
  if (!c || !p || !n) return SH_ERR_ARG;
  *p = buf<void>(c, "landmarks");
  *n = (size_t)c->B * sizeof(sh_landmarks);
  return *p ? SH_OK : SH_ERR_STATE;
}

// ---- parameters ------------------------------------------------------------------------------------
static int upload_params(sh_ctx* c) {
  c->packed_kind = -1; c->packed_x3 = false; c->packed_rfc = false;
  const size_t N = c->h_feat.size(), T = c->h_roots.size();
  const size_t bytes = c->unet_floats * 4 + N * 4 * 5 + T * 4;
  int rc = ensure(c, "params", bytes ? bytes : 16, 4);
  if (rc != SH_OK) return rc;
  char* p = buf<char>(c, "params");
  size_t o = 0;
  auto put = [&](const void* src, size_t n) -> hipError_t {
    hipError_t e = n ? hipMemcpyAsync(p + o, src, n, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    o += n;
    return e;
  };
  HIPCHK(c, put(c->h_unet.data(), c->unet_floats * 4));
  HIPCHK(c, put(c->h_feat.data(), N * 4));
  HIPCHK(c, put(c->h_thr.data(), N * 4));
  HIPCHK(c, put(c->h_ti.data(), N * 4));
  HIPCHK(c, put(c->h_fi.data(), N * 4));
  HIPCHK(c, put(c->h_lw.data(), N * 4));
  HIPCHK(c, put(c->h_roots.data(), T * 4));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SH_OK;
}

int sh_load_rfc(sh_ctx* c, const int32_t* feat, const float* thr, const int32_t* ti, const int32_t* fi, const float* lw, int n_nodes,
                const int32_t* roots, int n_trees) {
  if (!c || !feat || !thr || !ti || !fi || !lw || !roots || n_nodes <= 0 || n_trees <= 0) return fail(c, SH_ERR_ARG, "sh_load_rfc: bad argument");
  for (int i = 0; i < n_nodes; ++i) {
    if (feat[i] < 0 || feat[i] >= 9) return fail(c, SH_ERR_ARG, "sh_load_rfc: feature id out of range");
    if ((ti[i] < 0) != (fi[i] < 0) || ti[i] >= n_nodes || fi[i] >= n_nodes) return fail(c, SH_ERR_ARG, "sh_load_rfc: bad child index");
  }
  for (int t = 0; t < n_trees; ++t) if (roots[t] < 0 || roots[t] >= n_nodes) return fail(c, SH_ERR_ARG, "sh_load_rfc: bad root");
  {      // the walk on the device loops until it meets a leaf: every node must be reached once at most (a forest, no cycle)
    std::vector<char> seen(n_nodes, 0);
    std::vector<int> stack;
    for (int t = 0; t < n_trees; ++t) {
      stack.push_back(roots[t]);
      while (!stack.empty()) {
        const int i = stack.back(); stack.pop_back();
        if (seen[i]) return fail(c, SH_ERR_ARG, "sh_load_rfc: the node tables do not describe a forest (a node is reachable twice)");
        seen[i] = 1;
        if (ti[i] >= 0) { stack.push_back(ti[i]); stack.push_back(fi[i]); }
      }
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  c->h_feat.assign(feat, feat + n_nodes); c->h_thr.assign(thr, thr + n_nodes); c->h_ti.assign(ti, ti + n_nodes);
  c->h_fi.assign(fi, fi + n_nodes); c->h_lw.assign(lw, lw + n_nodes); c->h_roots.assign(roots, roots + n_trees);
  c->rfc_nodes = n_nodes; c->rfc_trees = n_trees; c->have_rfc = true; c->packed_rfc = false;
  return upload_params(c);
}

int sh_load_unet(sh_ctx* c, int base, int depth, const float* packed, size_t n_floats) {
  if (!c || !packed || depth < 1 || depth > 6 || base < 32 || base % 32 != 0 || base > SH_UNET_MAXBASE)
    return fail(c, SH_ERR_ARG, "sh_load_unet: bad argument (base must be a multiple of 32, at most 256; depth 1..6)");
  HIPCHK(c, hipSetDevice(c->device));
  // The layer table is built aside and swapped in together with the parameters only after the size check: a rejected
  // call leaves the loaded network (table, host copy, device block) as it was.
  std::map<std::string, sh_ctx::ULayer> layers;
  size_t o = 0;
  auto add = [&](const std::string& name, int taps, int cin, int cout) {
    sh_ctx::ULayer L;
    L.taps = taps; L.cin = cin; L.cout = cout;
    L.w_off = o; o += (size_t)taps * cin * cout;
    L.b_off = o; o += cout;
    layers[name] = L;
  };
  std::vector<int> ch(depth + 1);
  for (int i = 0; i <= depth; ++i) ch[i] = base << i;
  int cin = 1;
  for (int i = 0; i < depth; ++i) {
    add("enc" + std::to_string(i) + "a", 9, cin, ch[i]);
    add("enc" + std::to_string(i) + "b", 9, ch[i], ch[i]);
    cin = ch[i];
  }
  add("bota", 9, ch[depth - 1], ch[depth]);
  add("botb", 9, ch[depth], ch[depth]);
  for (int i = depth - 1; i >= 0; --i) {
    add("up" + std::to_string(i), 4, ch[i + 1], ch[i]);
    add("dec" + std::to_string(i) + "a", 9, 2 * ch[i], ch[i]);
    add("dec" + std::to_string(i) + "b", 9, ch[i], ch[i]);
  }
  { sh_ctx::ULayer L; L.taps = 1; L.cin = ch[0]; L.cout = 1; L.w_off = o; o += ch[0]; L.b_off = o; o += 1; layers["head"] = L; }
  if (o != n_floats) {
    char m[160];
    snprintf(m, sizeof m, "sh_load_unet: expected %zu floats for base=%d depth=%d, got %zu", o, base, depth, n_floats);
    return fail(c, SH_ERR_ARG, m);
  }
  for (size_t i = 0; i < n_floats; ++i)
    if (!std::isfinite(packed[i])) return fail(c, SH_ERR_ARG, "sh_load_unet: NaN / infinite parameter");
  (void)hipStreamSynchronize(c->stream);      // no forward of the previous network is still reading the block
  c->ulayers.swap(layers);
  c->packtab_ready = false;
  c->h_unet.assign(packed, packed + n_floats);
  c->unet_floats = n_floats; c->unet_base = base; c->unet_depth = depth; c->have_unet = true;
  int rc = upload_params(c);
  if (rc != SH_OK) { c->have_unet = false; c->ulayers.clear(); }
  return rc;
}

int sh_param_block_commit(sh_ctx* c) {
  if (!c) return SH_ERR_ARG;
  c->packed_kind = -1; c->packed_x3 = false; c->packed_rfc = false;
  auto it = c->bufs.find("params");
  if (it == c->bufs.end()) return fail(c, SH_ERR_STATE, "sh_param_block_commit: no parameters loaded");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t N = c->h_feat.size(), T = c->h_roots.size();
  std::vector<float> unet(c->unet_floats), thr(N), lw(N);
  std::vector<int32_t> feat(N), ti(N), fi(N), roots(T);
  const char* p = (const char*)it->second.p;
  size_t o = 0;
  auto get = [&](void* dst, size_t n) -> hipError_t {
    hipError_t e = n ? hipMemcpyAsync(dst, p + o, n, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    o += n;
    return e;
  };
  HIPCHK(c, get(unet.data(), c->unet_floats * 4));
  HIPCHK(c, get(feat.data(), N * 4));
  HIPCHK(c, get(thr.data(), N * 4));
  HIPCHK(c, get(ti.data(), N * 4));
  HIPCHK(c, get(fi.data(), N * 4));
  HIPCHK(c, get(lw.data(), N * 4));
  HIPCHK(c, get(roots.data(), T * 4));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the device walks the forest until it meets a leaf: what arrived must still be a forest over the same node count
  for (size_t i = 0; i < N; ++i) {
    if (feat[i] < 0 || feat[i] >= 9) return fail(c, SH_ERR_ARG, "sh_param_block_commit: feature id out of range in the device block");
    if ((ti[i] < 0) != (fi[i] < 0) || ti[i] >= (int)N || fi[i] >= (int)N) return fail(c, SH_ERR_ARG, "sh_param_block_commit: bad child index in the device block");
  }
  {
    std::vector<char> seen(N, 0);
    std::vector<int> stack;
    for (size_t t = 0; t < T; ++t) {
      if (roots[t] < 0 || roots[t] >= (int)N) return fail(c, SH_ERR_ARG, "sh_param_block_commit: bad root in the device block");
      stack.push_back(roots[t]);
      while (!stack.empty()) {
        const int i = stack.back(); stack.pop_back();
        if (seen[i]) return fail(c, SH_ERR_ARG, "sh_param_block_commit: the device block does not describe a forest");
        seen[i] = 1;
        if (ti[i] >= 0) { stack.push_back(ti[i]); stack.push_back(fi[i]); }
      }
    }
  }
  c->h_unet.swap(unet); c->h_feat.swap(feat); c->h_thr.swap(thr); c->h_ti.swap(ti); c->h_fi.swap(fi); c->h_lw.swap(lw); c->h_roots.swap(roots);
  return SH_OK;
}

int sh_param_block(sh_ctx* c, void** p, size_t* n) {
  if (!c || !p || !n) return SH_ERR_ARG;
  c->packed_kind = -1; c->packed_x3 = false; c->packed_rfc = false;      // the caller may write the block from here on (and confirms with sh_param_block_commit)
  auto it = c->bufs.find("params");
  if (it == c->bufs.end()) return fail(c, SH_ERR_STATE, "sh_param_block: no parameters loaded");
  *p = it->second.p;
  *n = c->unet_floats * 4 + c->h_feat.size() * 20 + c->h_roots.size() * 4;
  return SH_OK;
}

#include "sh_comm.h"

}  // extern "C"
