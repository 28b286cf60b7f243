// sh_comm.h -- the three collective entry points of the C-ABI: one HOST PROCESS drives several GPUs, one sh_ctx
// per device, and shards a cohort of humeri over them.  The humeri are independent (the reference runs one `Humerus(stl)` at a
// time, bone.py:110-131), so the data path has no exchange step; what crosses the xGMI links is
//   * the parameter block, once (the reference loads the same pickled forest / ONNX blob in every process: bicipital_groove.py:21-25,
//     anatomic_neck.py:56-60), and
//   * the landmark records, once per step, to the context that reports them.
// RCCL is bound at run time (dlopen of librccl.so.1 on the first sh_comm_init_all): a host that runs one GPU, and the CPU-side
// symbol check of tests/test_abi_exports.py, never load it.  The multi-PROCESS launch (one rank per GPU under torch.distributed,
// bench.py --gpus N) does the same two transfers over torch's RCCL binding: shoulder_amd/dist.py.
// Included by shoulder_hip.hip inside its extern "C" block, after the context and the record emitters.
#pragma once
// (shoulder_hip.hip includes <dlfcn.h> and <rccl/rccl.h> at file scope -- the latter for its TYPES only: every call goes through
// the table below)

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("SHOULDER_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !n[0]) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      api.why = dlerror();
    }
    if (!api.lib) return;
    bool ok = true;
    auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) { ok = false; api.why = std::string("librccl: no symbol ") + s; } return p; };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.lib); api.lib = nullptr; }
  });
  return api.lib ? &api : nullptr;
}

#define RCCLCHK(ctx, api, expr)                                                                         \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return fail(ctx, SH_ERR_HIP, std::string(#expr ": ") + (api)->GetErrorString(r_)); \
  } while (0)

// the contexts of one group, in rank order, as sh_comm_init_all left them
static int comm_group_check(sh_ctx** ctxs, int n, const char* who) {
  if (!ctxs || n < 1) return SH_ERR_ARG;
  for (int i = 0; i < n; ++i) if (!ctxs[i]) return SH_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    sh_ctx* c = ctxs[i];
    if (!c->comm || c->comm_n != n || c->comm_rank != i)
      return fail(ctxs[0], SH_ERR_STATE, std::string(who) + ": the contexts are not the group sh_comm_init_all made (same contexts, same order)");
    if (c->n_pending != 0) return fail(ctxs[0], SH_ERR_STATE, std::string(who) + ": runs are in flight (sh_collect them first)");
  }
  return SH_OK;
}

static void comm_forget(sh_ctx* c) {
  if (!c->comm) return;
  if (RcclApi* api = rccl_api()) { (void)hipSetDevice(c->device); (void)api->CommDestroy((ncclComm_t)c->comm); }
  c->comm = nullptr; c->comm_rank = -1; c->comm_n = 0;
}

// One communicator per context, rank i = ctxs[i] (ncclCommInitAll over the contexts' devices).  The contexts sit on DISTINCT
// devices; a context leaves its group when it is destroyed or joins another.  Errors are left on ctxs[0].
int sh_comm_init_all(sh_ctx** ctxs, int n) {
  if (!ctxs || n < 1) return SH_ERR_ARG;
  for (int i = 0; i < n; ++i) if (!ctxs[i]) return SH_ERR_ARG;
  sh_ctx* c0 = ctxs[0];
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return fail(c0, SH_ERR_ARG, "sh_comm_init_all: two contexts on the same device (one rank per GPU; lanes of one GPU share their records on the device)");
    if (ctxs[i]->n_pending != 0) return fail(c0, SH_ERR_STATE, "sh_comm_init_all: runs are in flight");
  }
  RcclApi* api = rccl_api();
  if (!api) return fail(c0, SH_ERR_STATE, "sh_comm_init_all: librccl could not be loaded (set SHOULDER_RCCL_LIB to its path)");
  for (int i = 0; i < n; ++i) comm_forget(ctxs[i]);
  std::vector<ncclComm_t> comms(n, nullptr);
  RCCLCHK(c0, api, api->CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_n = n; }
  return SH_OK;
}

// The parameter block of ctxs[root] (UNet parameters + forest tables, the layout of sh_param_block) to every context of the group.
// Every context has loaded a network and a forest of the SAME SHAPE before (sh_load_unet / sh_load_rfc with any values: the block
// is sized by them); the receivers check what arrived as sh_param_block_commit does.
int sh_bcast_weights(sh_ctx** ctxs, int n, int root) {
  int rc = comm_group_check(ctxs, n, "sh_bcast_weights");
  if (rc != SH_OK) return rc;
  sh_ctx* c0 = ctxs[0];
  if (root < 0 || root >= n) return fail(c0, SH_ERR_ARG, "sh_bcast_weights: root out of range");
  RcclApi* api = rccl_api();
  std::vector<void*> blk(n);
  size_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    size_t nb = 0;
    if ((rc = sh_param_block(ctxs[i], &blk[i], &nb)) != SH_OK) return fail(c0, rc, "sh_bcast_weights: a context has no parameters loaded");
    if (i == 0) bytes = nb;
    const sh_ctx *a = ctxs[i], *r = ctxs[root];
    if (nb != bytes || a->unet_base != r->unet_base || a->unet_depth != r->unet_depth || a->rfc_nodes != r->rfc_nodes || a->rfc_trees != r->rfc_trees)
      return fail(c0, SH_ERR_STATE, "sh_bcast_weights: the contexts hold parameter blocks of different shapes");
    HIPCHK(c0, hipSetDevice(a->device));
    HIPCHK(c0, hipStreamSynchronize(a->stream));
  }
  RCCLCHK(c0, api, api->GroupStart());
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(ctxs[i]->device);
    ncclResult_t r = api->Broadcast(blk[i], blk[i], bytes, ncclChar, root, (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream);
    if (r != ncclSuccess) { (void)api->GroupEnd(); return fail(c0, SH_ERR_HIP, std::string("ncclBroadcast: ") + api->GetErrorString(r)); }
  }
  RCCLCHK(c0, api, api->GroupEnd());
  for (int i = 0; i < n; ++i) {
    HIPCHK(c0, hipSetDevice(ctxs[i]->device));
    HIPCHK(c0, hipStreamSynchronize(ctxs[i]->stream));
  }
  for (int i = 0; i < n; ++i) {
    if (i == root) continue;
    if ((rc = sh_param_block_commit(ctxs[i])) != SH_OK) return fail(c0, rc, std::string("sh_bcast_weights: rank ") + std::to_string(i) + ": " + ctxs[i]->err);
  }
  return SH_OK;
}

// The records of every context's LAST run, in rank order, into host memory at ctxs[0]: sum of the batch sizes records, each
// sh_record_bytes(rows) long, `rows` being the record format every context of the group is set to (sh_set_record_rows; 0 = full
// sh_landmarks records).  Ranks send from their device buffers to rank 0's (ncclSend / ncclRecv, one group), rank 0 copies out once.
int sh_gather_landmarks(sh_ctx** ctxs, int n, sh_landmarks* out_root) {
  int rc = comm_group_check(ctxs, n, "sh_gather_landmarks");
  if (rc != SH_OK) return rc;
  sh_ctx* c0 = ctxs[0];
  if (!out_root) return fail(c0, SH_ERR_ARG, "sh_gather_landmarks: null output");
  RcclApi* api = rccl_api();
  const int rows = c0->rec_rows;
  const size_t rec = rec_bytes_rows(rows);
  long long total = 0;
  std::vector<long long> off(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    sh_ctx* c = ctxs[i];
    if (c->rec_rows != rows) return fail(c0, SH_ERR_STATE, "sh_gather_landmarks: the contexts are set to different record formats (sh_set_record_rows)");
    if (c->B <= 0 || c->bufs.find("landmarks") == c->bufs.end() || !c->bufs["landmarks"].p)
      return fail(c0, SH_ERR_STATE, "sh_gather_landmarks: a context has no run to report");
    total += c->B;
    off[i + 1] = total;
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  if ((rc = ensure(c0, "comm.gather", (size_t)total * rec, 1)) != SH_OK) return rc;
  char* gather = (char*)c0->bufs["comm.gather"].p;
  std::vector<const void*> send(n, nullptr);
  for (int i = 0; i < n; ++i) {
    sh_ctx* c = ctxs[i];
    HIPCHK(c0, hipSetDevice(c->device));
    c->b0 = 0; c->Bwin = c->B;
    if (i == 0) { if ((rc = emit_records(c, gather, 0, c->B, rows, rec)) != SH_OK) return fail(c0, rc, c->err); }
    else if (rows <= 0) send[i] = c->bufs["landmarks"].p;      // full records leave from where the run wrote them
    else {
      if ((rc = ensure(c, "comm.send", (size_t)c->B * rec, 1)) != SH_OK) return fail(c0, rc, c->err);
      if ((rc = emit_records(c, c->bufs["comm.send"].p, 0, c->B, rows, rec)) != SH_OK) return fail(c0, rc, c->err);
      send[i] = c->bufs["comm.send"].p;
    }
  }
  if (n > 1) {
    RCCLCHK(c0, api, api->GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int i = 1; i < n && r == ncclSuccess; ++i) {
      (void)hipSetDevice(ctxs[i]->device);
      r = api->Send(send[i], (size_t)ctxs[i]->B * rec, ncclChar, 0, (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream);
      if (r != ncclSuccess) break;
      (void)hipSetDevice(c0->device);
      r = api->Recv(gather + (size_t)off[i] * rec, (size_t)ctxs[i]->B * rec, ncclChar, i, (ncclComm_t)c0->comm, c0->stream);
    }
    if (r != ncclSuccess) { (void)api->GroupEnd(); return fail(c0, SH_ERR_HIP, std::string("ncclSend / ncclRecv: ") + api->GetErrorString(r)); }
    RCCLCHK(c0, api, api->GroupEnd());
  }
  HIPCHK(c0, hipSetDevice(c0->device));
  HIPCHK(c0, hipMemcpyAsync(out_root, gather, (size_t)total * rec, hipMemcpyDeviceToHost, c0->stream));
  for (int i = n - 1; i >= 0; --i) {
    HIPCHK(c0, hipSetDevice(ctxs[i]->device));
    HIPCHK(c0, hipStreamSynchronize(ctxs[i]->stream));
  }
  return SH_OK;
}
