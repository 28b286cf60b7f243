// shoulder_hip.hip -- libshoulder_hip.so: context, buffers, C-ABI (include/shoulder_hip.h) and the
// stage runner.  gfx950 only.  Kernels live in k_*.h next to this file.
#include "../../include/shoulder_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "k_slices.h"
#include "k_stages.h"

using namespace sh;

// ---------------------------------------------------------------------------------------------
struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
  int elem = 1;
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
  bool obb_injected = false;
  // timing
  bool timing = false;
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
  return it == c->bufs.end() ? nullptr : (T*)it->second.p;
}

// kernel launch with optional HIP-event timing on the ctx stream
#define LAUNCH(ctx, name, kernel, grid, block, ...)                                         \
  do {                                                                                      \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                \
    if ((ctx)->timing) {                                                                    \
      (void)hipEventCreate(&e0_); (void)hipEventCreate(&e1_);                               \
      (void)hipEventRecord(e0_, (ctx)->stream);                                             \
    }                                                                                       \
    hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);                 \
    if ((ctx)->timing) {                                                                    \
      (void)hipEventRecord(e1_, (ctx)->stream);                                             \
      (ctx)->pending.emplace_back(name, e0_, e1_);                                          \
    }                                                                                       \
    HIPCHK(ctx, hipGetLastError());                                                         \
  } while (0)

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
  p->pad_ = 0;
  return SH_OK;
}

int sh_ctx_create(int device, void* hip_stream, sh_ctx** out) {
  if (!out) return SH_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SH_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return SH_ERR_HIP;
  sh_ctx* c = new (std::nothrow) sh_ctx();
  if (!c) return SH_ERR_NOMEM;
  c->device = device;
  sh_default_params(&c->params);
  if (hip_stream) c->stream = (hipStream_t)hip_stream;
  else {
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return SH_ERR_HIP; }
    c->own_stream = true;
  }
  *out = c;
  return SH_OK;
}

void sh_ctx_destroy(sh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  drain_timers(c);
  for (auto& kv : c->bufs)
    if (kv.second.p) (void)hipFree(kv.second.p);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* sh_last_error(const sh_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int sh_set_params(sh_ctx* c, const sh_params* p) {
  if (!c || !p) return SH_ERR_ARG;
  int a, b;
  cutoff_range(SH_NPROX, p->groove_cutoff[0], p->groove_cutoff[1], &a, &b);
  if (b - a != SH_GROOVE_NROWS) return fail(c, SH_ERR_ARG, "groove_cutoff must select 330 proximal rows");
  cutoff_range(SH_NFULL, p->canal_cutoff[0], p->canal_cutoff[1], &a, &b);
  if (b - a < 2 || a < 0 || b > SH_NFULL) return fail(c, SH_ERR_ARG, "canal_cutoff selects fewer than 2 slices");
  c->params = *p;
  return SH_OK;
}

int sh_batch_size(const sh_ctx* c) { return c ? c->B : 0; }

// ---- meshes ------------------------------------------------------------------------------------
static int alloc_batch(sh_ctx* c) {
  const int B = c->B;
  int rc;
#define ENS(name, bytes, elem) if ((rc = ensure(c, name, (size_t)(bytes), elem)) != SH_OK) return rc
  ENS("verts_obb", c->sumV * 3 * 8, 8);
  ENS("obb_transform", B * 16 * 8, 8);
  ENS("zb_enc", B * 2 * 8, 8);
  ENS("z_bounds", B * 2 * 8, 8);
  ENS("z_length", B * 8, 8);
  ENS("err", B * 4, 4);
  ENS("neck_z", B * 8, 8);
  ENS("neck_index", B * 4, 4);
  ENS("cpd_scratch", (size_t)B * 6144 * 8, 8);
  ENS("canal.points_obb", B * 80 * 3 * 8, 8);
  ENS("canal.axis_obb", B * 6 * 8, 8);
  ENS("canal.axis_ct", B * 6 * 8, 8);
  ENS("landmarks", (size_t)B * sizeof(sh_landmarks), 1);
  struct S { const char* p; int N; bool ring; };
  const S sets[3] = {{"full", SH_NFULL, false}, {"distal", SH_NDIST, true}, {"prox", SH_NPROX, true}};
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
  ENS("prox.ixy", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
  ENS("prox.itr_start", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
  ENS("prox.itr_centered_start", (size_t)B * SH_NPROX * 2 * SH_MPROX * 8, 8);
#undef ENS
  c->obb_injected = false;
  return SH_OK;
}

int sh_upload_meshes(sh_ctx* c, const float* verts, const int32_t* faces, const int64_t* v_off, const int64_t* f_off, int B) {
  if (!c || !verts || !faces || !v_off || !f_off || B <= 0) return fail(c, SH_ERR_ARG, "sh_upload_meshes: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->h_voff.assign(v_off, v_off + B + 1);
  c->h_foff.assign(f_off, f_off + B + 1);
  c->sumV = v_off[B]; c->sumF = f_off[B];
  c->maxV = c->maxF = 0;
  for (int b = 0; b < B; ++b) {
    long long nv = v_off[b + 1] - v_off[b], nf = f_off[b + 1] - f_off[b];
    if (nv < 4 || nf < 4) return fail(c, SH_ERR_ARG, "sh_upload_meshes: a mesh has fewer than 4 vertices/faces");
    c->maxV = std::max(c->maxV, nv); c->maxF = std::max(c->maxF, nf);
    for (long long i = 3 * f_off[b]; i < 3 * f_off[b + 1]; ++i)
      if (faces[i] < 0 || faces[i] >= nv) return fail(c, SH_ERR_ARG, "sh_upload_meshes: face index out of range");
  }
  c->B = B;
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
  return alloc_batch(c);
}

int sh_synth_batch(sh_ctx* c, const double* T, int B) {
  if (!c || !T || B <= 0) return fail(c, SH_ERR_ARG, "sh_synth_batch: bad argument");
  if (c->B < 1) return fail(c, SH_ERR_STATE, "sh_synth_batch: upload a template mesh first");
  HIPCHK(c, hipSetDevice(c->device));
  const long long V = c->h_voff[1] - c->h_voff[0], F = c->h_foff[1] - c->h_foff[0];
  // keep the template aside
  int rc;
  if ((rc = ensure(c, "tmpl_verts", V * 3 * 4, 4)) != SH_OK) return rc;
  if ((rc = ensure(c, "tmpl_faces", F * 3 * 4, 4)) != SH_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(buf<float>(c, "tmpl_verts"), buf<float>(c, "verts") + 3 * c->h_voff[0], V * 3 * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(buf<int>(c, "tmpl_faces"), buf<int>(c, "faces") + 3 * c->h_foff[0], F * 3 * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = B;
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
  return alloc_batch(c);
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
  c->timing = on != 0;
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

// ---- stage runner ----------------------------------------------------------------------------------
static int run_slice_set(sh_ctx* c, const char* pfx, int kind, int N, bool ring, bool resample) {
  const int B = c->B;
  std::string p = pfx;
  double* zs = buf<double>(c, (p + ".zs").c_str());
  double* zeff = buf<double>(c, (p + ".zeff").c_str());
  int* cnt = buf<int>(c, (p + ".seg_count").c_str());
  Seg* segs = buf<Seg>(c, (p + ".segs").c_str());
  LAUNCH(c, "k_make_planes", k_make_planes, dim3(B), dim3(256), kind, N, buf<double>(c, "z_bounds"), buf<double>(c, "neck_z"), zs, zeff, B);
  HIPCHK(c, hipMemsetAsync(cnt, 0, (size_t)B * N * 4, c->stream));
  dim3 g((unsigned)std::min<long long>((c->maxF + 255) / 256, 4096), (unsigned)B);
  LAUNCH(c, "k_slice_emit", k_slice_emit, g, dim3(256), buf<double>(c, "verts_obb"), buf<int>(c, "faces"), buf<long long>(c, "voff"),
         buf<long long>(c, "foff"), zeff, N, cnt, segs, buf<int>(c, "err"));
  LAUNCH(c, "k_slice_link", k_slice_link, dim3(B * N), dim3(SH_LINK_THREADS), N, cnt, segs, buf<double>(c, (p + ".centroids").c_str()),
         buf<double>(c, (p + ".areas").c_str()), buf<int>(c, (p + ".nloops").c_str()), buf<int>(c, (p + ".ring_n").c_str()),
         ring ? buf<double>(c, (p + ".ring").c_str()) : (double*)nullptr, 0, buf<int>(c, "err"));
  if (resample) {
    LAUNCH(c, "k_resample_polar", k_resample_polar, dim3(B * N), dim3(SH_RS_THREADS), N, SH_MPROX, buf<int>(c, (p + ".ring_n").c_str()),
           buf<double>(c, (p + ".ring").c_str()), buf<double>(c, (p + ".centroids").c_str()), buf<double>(c, "prox.ixy"),
           buf<double>(c, "prox.itr_start"), buf<double>(c, "prox.itr_centered_start"));
  }
  return SH_OK;
}

int sh_run(sh_ctx* c, uint32_t mask, sh_landmarks* out) {
  if (!c) return SH_ERR_ARG;
  if (c->B < 1) return fail(c, SH_ERR_STATE, "sh_run: no meshes uploaded");
  HIPCHK(c, hipSetDevice(c->device));
  const int B = c->B;
  int rc;
  HIPCHK(c, hipMemsetAsync(buf<int>(c, "err"), 0, B * 4, c->stream));
  if (mask & SH_STAGE_OBB) {
    return fail(c, SH_ERR_STATE, "sh_run: SH_STAGE_OBB not available in this build");
  } else if (!c->obb_injected) {
    return fail(c, SH_ERR_STATE, "sh_run: no OBB transform (run SH_STAGE_OBB or sh_store(\"obb_transform\"))");
  }
  if (mask & (SH_STAGE_OBB | SH_STAGE_FULL)) {
    // verts_obb + z bounds (mesh.py:85-86)
    HIPCHK(c, hipMemsetAsync(buf<unsigned long long>(c, "zb_enc"), 0, 0, c->stream));
    LAUNCH(c, "k_init_bounds", k_init_bounds, dim3((2 * B + 63) / 64), dim3(64), buf<unsigned long long>(c, "zb_enc"), B);
    dim3 g((unsigned)std::min<long long>((c->maxV + 255) / 256, 1024), (unsigned)B);
    LAUNCH(c, "k_transform_verts", k_transform_verts, g, dim3(256), buf<float>(c, "verts"), buf<long long>(c, "voff"),
           buf<double>(c, "obb_transform"), buf<double>(c, "verts_obb"), buf<unsigned long long>(c, "zb_enc"));
    LAUNCH(c, "k_decode_bounds", k_decode_bounds, dim3((2 * B + 63) / 64), dim3(64), buf<unsigned long long>(c, "zb_enc"),
           buf<double>(c, "z_bounds"), B);
  }
  if (mask & SH_STAGE_FULL)
    if ((rc = run_slice_set(c, "full", 0, SH_NFULL, false, false)) != SH_OK) return rc;
  if (mask & SH_STAGE_DISTAL)
    if ((rc = run_slice_set(c, "distal", 2, SH_NDIST, true, false)) != SH_OK) return rc;
  if (mask & SH_STAGE_NECK) {
    LAUNCH(c, "k_neck", k_neck, dim3((B + 63) / 64), dim3(64), buf<double>(c, "full.areas"), buf<double>(c, "full.zs"),
           buf<double>(c, "cpd_scratch"), buf<double>(c, "neck_z"), buf<int>(c, "neck_index"), B);
  }
  if (mask & SH_STAGE_CANAL) {
    LAUNCH(c, "k_canal", k_canal, dim3(B), dim3(64), buf<double>(c, "full.centroids"), buf<double>(c, "full.zs"),
           buf<double>(c, "z_bounds"), buf<double>(c, "obb_transform"), c->params.canal_cutoff[0], c->params.canal_cutoff[1],
           buf<double>(c, "canal.points_obb"), buf<double>(c, "canal.axis_obb"), buf<double>(c, "canal.axis_ct"));
  }
  if (mask & SH_STAGE_PROXIMAL)
    if ((rc = run_slice_set(c, "prox", 1, SH_NPROX, true, true)) != SH_OK) return rc;
  if (mask & (SH_STAGE_GROOVE | SH_STAGE_ANP | SH_STAGE_TE | SH_STAGE_CSYS))
    return fail(c, SH_ERR_STATE, "sh_run: groove/ANP/TE/csys stages not available in this build");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int> herr(B);
  HIPCHK(c, hipMemcpy(herr.data(), buf<int>(c, "err"), B * 4, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b)
    if (herr[b] != 0) {
      char m[128];
      snprintf(m, sizeof m, "mesh %d: device stage error %d (capacity=-4, geometry=-5)", b, herr[b]);
      return fail(c, herr[b], m);
    }
  (void)out;
  return SH_OK;
}

int sh_landmarks_device(sh_ctx* c, void** p, size_t* n) {
  if (!c || !p || !n) return SH_ERR_ARG;
  *p = buf<void>(c, "landmarks");
  *n = (size_t)c->B * sizeof(sh_landmarks);
  return *p ? SH_OK : SH_ERR_STATE;
}

int sh_load_rfc(sh_ctx* c, const int32_t*, const float*, const int32_t*, const int32_t*, const float*, int, const int32_t*, int) {
  return fail(c, SH_ERR_STATE, "sh_load_rfc: not available in this build");
}
int sh_load_unet(sh_ctx* c, int, int, const float*, size_t) { return fail(c, SH_ERR_STATE, "sh_load_unet: not available in this build"); }
int sh_param_block(sh_ctx* c, void**, size_t*) { return fail(c, SH_ERR_STATE, "sh_param_block: not available in this build"); }

}  // extern "C"
