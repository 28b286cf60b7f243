/* shoulder_hip.h -- C-ABI of libshoulder_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the `shoulder.Humerus` landmark path of gregspangenberg/shoulder.
 * The reference has no FFI of its own (pure Python over NumPy/trimesh/onnxruntime); the
 * Python facade `shoulder_amd.Humerus` keeps the reference's accessor API and calls these
 * entry points through ctypes.  Each entry point names the reference code it replaces
 * (paths relative to src/shoulder/).
 *
 * Conventions: every function returns 0 (SH_OK) or a negative sh_status and never throws or
 * aborts across the boundary; sh_last_error() gives the text.  The caller owns every host
 * buffer; the library owns all device memory inside sh_ctx.  One sh_ctx per HIP device /
 * stream; a ctx is NOT thread-safe; distinct ctxs may run concurrently.  All work is
 * enqueued on the ctx's HIP stream; functions that return host data synchronise that stream.
 * Matrices are 4x4 row-major float64, points are xyz float64, vertices are float32 (as in
 * an STL file).
 */
#ifndef SHOULDER_HIP_H
#define SHOULDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sh_ctx sh_ctx;

typedef enum sh_status {
  SH_OK = 0,
  SH_ERR_ARG = -1,      /* bad argument / shape */
  SH_ERR_HIP = -2,      /* HIP runtime error (text in sh_last_error) */
  SH_ERR_STATE = -3,    /* call order: meshes / parameters not loaded */
  SH_ERR_CAPACITY = -4, /* a capacity was exceeded.  Since round 5 the stages grow what they need (crossings and loops per
                           section, end-section points, hull vertices / faces, silhouette edges of a box candidate: the run
                           is repeated inside sh_collect) -- what is left: a caller-sized output that is too small (the
                           counts returned say how large), more than 1 024 closed loops in ONE section, and a growth
                           that is needed while a second run is in flight (collect it, run again) */
  SH_ERR_GEOMETRY = -5, /* degenerate input: open contour, empty slice, ray miss, ... */
  SH_ERR_NOMEM = -6
} sh_status;

/* Stages of sh_run, in evaluation order (bone.py:110-157). */
enum {
  SH_STAGE_OBB      = 1u << 0, /* mesh.py:63-125  FullObb._obb (hull, min-volume box, head-end flip) */
  SH_STAGE_FULL     = 1u << 1, /* slice.py:209-224 + :21-60  FullSlices sections, centroids, areas   */
  SH_STAGE_NECK     = 1u << 2, /* surgical_neck.py:22-56  kernel change point -> neck_z              */
  SH_STAGE_CANAL    = 1u << 3, /* canal.py:19-85                                                     */
  SH_STAGE_PROXIMAL = 1u << 4, /* slice.py:227-253 + :65-147 ProximalSlices, resample, polar images  */
  SH_STAGE_GROOVE   = 1u << 5, /* bicipital_groove.py:26-265                                         */
  SH_STAGE_ANP      = 1u << 6, /* anatomic_neck.py:31-236 (image, UNet, edge points, plane, axes)    */
  SH_STAGE_DISTAL   = 1u << 7, /* slice.py:256-276 DistalSlices sections                             */
  SH_STAGE_TE       = 1u << 8, /* epicondyle.py:29-101                                               */
  SH_STAGE_CSYS     = 1u << 9, /* bone.py:146-157 construct_csys + re-expression of landmarks        */
  SH_STAGE_APPLY    = 1u << 10,/* bone.py:155 `mesh_ct.copy().apply_transform(csys)` for the whole batch: device buffer
                                  "verts_csys" (sumV x 3 float64) = csys[b] * vertices of mesh b; needs SH_STAGE_CSYS in
                                  the same run                                                        */
  SH_STAGE_ALL      = 0x7FFu
};

/* Arithmetic of the anatomic-neck network (the ONNX session of anatomic_neck.py:62-76 computes in float32):
 * F32  one float32 fma chain per output on v_mfma_f32_16x16x4_f32 -- bit-exact against oracle/unet_chain.c;
 * BF16 / F16  16-bit activations and weights on v_mfma_f32_16x16x32_{bf16,f16}, float32 accumulate (throughput paths;
 *      F16 carries 11 significant bits instead of 8 and needs activations below 65504). */
enum { SH_UNET_F32 = 0, SH_UNET_BF16 = 1, SH_UNET_F16 = 2,
       SH_UNET_F32X = 3 };      /* f32 tensors, MFMA layers on split f16 operands (3 MFMAs per product): f32-grade logits at ~5x the f32 rate.
                                 * Range (the operands are pairs of f16 values): weights of the >= 32-channel layers need |w| < 1023.5
                                 * (= 65504 / 64; a run with a larger weight returns SH_ERR_ARG and names the layer), activations need
                                 * |x| < 65504 (beyond it an operand's high part is an infinity and the logits NaN, as in F16), and the
                                 * low part of an activation below 2^-14 falls into f16's subnormals (absolute error <= 2^-25).  The
                                 * reference's input is an image in [0, 1] (anatomic_neck.py:57); SH_UNET_F32 has none of these limits. */
/* Which facade class of bone.py the meshes are: `Humerus` (bone.py:110-157) or `ProximalHumerus` (bone.py:24-64: a
 * humerus cut in the shaft -- ProxObb head-end rule and canal range mesh.py:128-192, neck cut-off (0.2, 0.99)
 * surgical_neck.py:25-26, canal cut-offs from the box canal.py:33-38, no distal / trans-epicondylar stage,
 * csys = apply_csys_canal_articular bone.py:53-62). */
enum { SH_BONE_HUMERUS = 0, SH_BONE_PROXIMAL = 1 };

#define SH_GROOVE_ROWS 330   /* rows 150..479 of 600 proximal slices (slice.py:157-164) */
#define SH_ANP_MAX_PTS 4096  /* capacity of the padded edge-point list                  */

/* Fixed-size result per humerus; everything in CT coordinates unless noted.
 * Row order of every (2,3) axis follows the reference accessor it mirrors. */
typedef struct sh_landmarks {
  double obb_transform[16];   /* FullObb.transform, CT -> OBB incl. head-end flip (mesh.py:124) */
  double z_length;            /* mesh.py:86 */
  double neck_z;              /* SurgicalNeck.neck_z, OBB frame (surgical_neck.py:34) */
  double canal_axis[6];       /* Canal.axis(): [proximal, distal] (canal.py:58-85) */
  double te_axis[6];          /* TransEpicondylar.axis(): [medial, lateral] (epicondyle.py:29-101) */
  double groove_axis[6];      /* DeepGroove.axis() (bicipital_groove.py:244-265) */
  double bg_theta;            /* DeepGroove.bg_theta (bicipital_groove.py:188) */
  double anp_plane_point[3];  /* AnatomicNeck.plane().point  (anatomic_neck.py:123-153) */
  double anp_plane_normal[3]; /* AnatomicNeck.plane().normal */
  double anp_axis_normal[6];  /* AnatomicNeck.axis_normal(): [upper, lower] (anatomic_neck.py:174-200) */
  double anp_axis_central[6]; /* AnatomicNeck.axis_central(): [upper, lower] (anatomic_neck.py:202-236) */
  double csys[16];            /* apply_csys_canal_transepiconylar() matrix, CT -> canal/TE (bone.py:146-157);
                                 SH_BONE_PROXIMAL: apply_csys_canal_articular() (bone.py:53-62) */
  double csys_articular[16];  /* apply_csys_canal_articular() matrix, CT -> canal / head-normal axis (bone.py:53-62), both bone kinds */
  double neckshaft;           /* NeckShaft.calc(), degrees (bone_props.py:88-112) */
  double retroversion;        /* RetroVersion.calc() with landmarks in CT, degrees (bone_props.py:50-85); NaN for SH_BONE_PROXIMAL */
  double radius_curvature;    /* RadiusCurvature.calc(), mm (bone_props.py:115-148) */
  double canal_cutoff[2];     /* cut-off fractions the canal used: sh_params.canal_cutoff, or ProxObb.cutoff_pcts (mesh.py:190) */
  double groove_points[SH_GROOVE_ROWS * 3]; /* DeepGroove.points() (bicipital_groove.py:26-242) */
  double anp_points[SH_ANP_MAX_PTS * 3];    /* AnatomicNeck.points(), first n_anp rows valid (anatomic_neck.py:31-121) */
  int32_t n_anp;              /* number of edge points (K) */
  int32_t n_articular;        /* number of mask pixels (anatomic_neck.py:104-112) */
  int32_t neck_index;         /* change-point index into areas1((0.70,0.99)) (surgical_neck.py:33) */
  int32_t flipped;            /* 1 if the head end was at -z of the raw box (mesh.py:112) */
  int32_t status;             /* 0 or a negative sh_status for this mesh */
  int32_t side;               /* Side.calc(): 0 = "left", 1 = "right" (bone_props.py:12-47) */
} sh_landmarks;

/* Tunables the reference exposes as keyword defaults (SURVEY 5 "config / flags"). */
typedef struct sh_params {
  double canal_cutoff[2];     /* canal.py:19          default (0.35, 0.75) */
  double groove_cutoff[2];    /* bicipital_groove.py:26 default (0.2, 0.75); rows must stay 330 */
  double groove_deg_window;   /* bicipital_groove.py:26 default 7 */
  int32_t unet_dtype;         /* SH_UNET_F32 (parity), SH_UNET_F32X (f32-grade, fast), SH_UNET_BF16 or SH_UNET_F16 (throughput) */
  int32_t bone_kind;          /* SH_BONE_HUMERUS (default) or SH_BONE_PROXIMAL */
} sh_params;

/* ---- context ------------------------------------------------------------------------ */
int  sh_ctx_create(int device, void* hip_stream /* nullable: e.g. torch's current stream */, sh_ctx** out);
void sh_ctx_destroy(sh_ctx*);
const char* sh_last_error(const sh_ctx*);      /* owned by ctx, valid until the next call */
int  sh_default_params(sh_params* out);
int  sh_set_params(sh_ctx*, const sh_params*);
int  sh_get_params(const sh_ctx*, sh_params* out);   /* the values in force (read-modify-write with sh_set_params) */

/* ---- parameters (replace the ONNX files read at bicipital_groove.py:174-180 and
 *      anatomic_neck.py:62-69; host pointers, copied) ---------------------------------- */
int  sh_load_rfc(sh_ctx*, const int32_t* feat, const float* thr, const int32_t* true_idx,
                 const int32_t* false_idx, const float* leaf_weight, int n_nodes,
                 const int32_t* roots, int n_trees);
/* UNet: `packed` = concatenation, in this order, of enc{i}a_w,enc{i}a_b,enc{i}b_w,enc{i}b_b (i=0..depth-1),
 * bota_w,bota_b,botb_w,botb_b, then for i=depth-1..0: up{i}_w,up{i}_b,dec{i}a_w,dec{i}a_b,dec{i}b_w,dec{i}b_b,
 * then head_w, head_b; conv weights laid out [ky][kx][cin][cout] float32.
 * base_channels: a multiple of 32, at most 256; depth 1..6. */
int  sh_load_unet(sh_ctx*, int base_channels, int depth, const float* packed, size_t n_floats);
/* Device address + size of the packed parameter block (UNet then RFC), for a collective
 * broadcast by the caller (torch.distributed over RCCL); valid until the next sh_load_*. */
int  sh_param_block(sh_ctx*, void** dev_ptr, size_t* nbytes);
/* After the caller has overwritten the device parameter block (the receiving ranks of a broadcast): re-read the host
 * mirrors from it, so that a later sh_load_rfc / sh_load_unet -- which re-uploads the whole block from the mirrors --
 * does not put stale values back.  The forest's topology (child indices, roots) is validated like in sh_load_rfc. */
int  sh_param_block_commit(sh_ctx*);
/* (The 16-bit UNet paths pack their weights from this block once and keep the packed copy until the block can have
 * changed: sh_load_*, sh_param_block / sh_buffer_device("params") handing the pointer out, sh_param_block_commit, sh_store.
 * A caller that keeps the pointer and writes the block again later must call sh_param_block_commit again.) */

/* ---- meshes (replace MeshLoader, mesh.py:14-41; vertices already merged) ------------- */
int  sh_upload_meshes(sh_ctx*, const float* verts /* sumV x 3 */, const int32_t* faces /* sumF x 3, per-mesh local ids */,
                      const int64_t* v_off /* B+1 */, const int64_t* f_off /* B+1 */, int B);
/* The same from B binary STL files held in host memory (`trimesh.load_mesh(stl_file)` at mesh.py:22-27 incl. its vertex
 * merge): records are parsed and merged on the device -- vertices with equal bit patterns (after -0.0 -> +0.0) become
 * one, numbered by first appearance in the file; triangles that use a vertex twice are dropped.  v_off_out / f_off_out
 * (B+1 each, nullable) receive the resulting offsets. */
int  sh_upload_stl(sh_ctx*, const void* const* files, const size_t* nbytes, int B, int64_t* v_off_out, int64_t* f_off_out);
/* A stream of NEW batches (the reference's unit of work is a new STL: mesh.py:22-27, bone.py:110-131).  sh_upload_* hands a
 * batch over synchronously; the staging calls hand the NEXT batch over while a run of the resident one executes:
 *   sh_stage_meshes / sh_stage_stl  same arguments and checks as sh_upload_meshes / sh_upload_stl.  The call checks sizes and
 *     headers and returns at once; a background thread copies the arrays / files through page-locked staging (page-locked
 *     caller memory -- sh_host_alloc -- is read in place) into buffers of their own on a copy stream, element checks and the STL
 *     parse / vertex merge run on the device, and the convex hulls of the staged batch (host hull mode) are computed by the same
 *     thread.  THE CALLER KEEPS THE ARRAYS / FILES UNCHANGED UNTIL sh_commit_staged HAS RETURNED (or the batch is replaced: staging
 *     again, any sh_upload_* / sh_synth_batch).  One batch can be staged at a time.
 *   sh_commit_staged  makes the staged batch the resident one (buffer entries are swapped, nothing is copied).  Needs the context
 *     idle (sh_collect every run first).  Reports what sh_upload_* would have reported for a bad batch (SH_ERR_ARG; the resident
 *     batch stays).  The next sh_submit / sh_run finds the hulls prepared.  v_off_out / f_off_out (B+1 each, nullable): offsets.
 * A run submitted between stage and commit belongs to the resident batch and voids the staged batch's prepared hulls (they are
 * computed again by its first run).  Keep sh_set_overlap off on a context that streams: a batch that runs once needs no hulls
 * prepared for a second run, and the staging call would wait for that preparation.  Records are identical to sh_upload_*
 * followed by the same runs. */
int  sh_stage_meshes(sh_ctx*, const float* verts, const int32_t* faces, const int64_t* v_off, const int64_t* f_off, int B);
int  sh_stage_stl(sh_ctx*, const void* const* files, const size_t* nbytes, int B);
int  sh_commit_staged(sh_ctx*, int64_t* v_off_out, int64_t* f_off_out);
int  sh_staged(const sh_ctx*);      /* 1 while a staged batch waits for its commit */
/* Synthetic batch (BASELINE config 3/4): mesh i = similarity transform T[i] (4x4, float64)
 * of uploaded mesh 0, evaluated on the device in float64 and stored as float32. */
int  sh_synth_batch(sh_ctx*, const double* T /* B x 16 */, int B);
int  sh_batch_size(const sh_ctx*);

/* The proximal slice set's resampling (slice.py:65-147, 166-206) makes three arrays per plane -- the 512-sample contour ("prox.ixy")
 * and its polar rows about the origin / about the centroid ("prox.itr_start" / "prox.itr_centered_start") -- 24 KB per plane, 944 MB
 * per batch of 64.  The stages behind read part of them only (polar rows about the origin from plane 88 on: anatomic_neck.py:34;
 * centred rows inside the groove's cut-off range: bicipital_groove.py:63-67; the contour never), and that is what a run writes.
 * sh_set_keep_products(ctx, 1): every plane's three arrays are written (for sh_fetch: the reference keeps them as
 * `Slices.ixy / itr_start / itr_centered_start`).  A run of SH_STAGE_GROOVE without SH_STAGE_PROXIMAL after groove_cutoff changed
 * returns SH_ERR_STATE: the rows it needs were not written. */
int  sh_set_keep_products(sh_ctx*, int on);

/* Records on the wire.  A full sh_landmarks record is 104 KB, 96 KB of it the padded anatomic-neck point list (4 096 rows; a
 * humerus has about a thousand).  sh_set_record_rows(R), R > 0: every record a run hands out through `out` of sh_run /
 * sh_submit (host memory or a gather's device send buffer) is PACKED to sh_record_bytes(R) = 8 680 + 24 R bytes:
 *   [the bytes of sh_landmarks in front of anp_points] [its six trailing int32 fields: n_anp, n_articular, neck_index,
 *   flipped, status, side] [R rows of anp_points: the first min(n_anp, R), zeros behind them]
 * n_anp keeps the true count; sh_anp_points returns every point of one humerus of the last run (CT, the record's own
 * arithmetic) whatever the format.  R = 0 (default): full records.  The device records (sh_landmarks_device) are always full. */
int  sh_set_record_rows(sh_ctx*, int anp_rows);
size_t sh_record_bytes(int anp_rows);
int  sh_anp_points(sh_ctx*, int b, double* out /* cap x 3 */, int cap, int* n_out);

/* ---- the hot path -------------------------------------------------------------------- */
int  sh_run(sh_ctx*, uint32_t stage_mask, sh_landmarks* out /* B, host, nullable */);
/* The same in two halves, for callers that stream runs: sh_submit enqueues a run (all device work, the copy of the records to
 * `out` -- page-locked memory, see sh_host_alloc, or the call blocks -- and of the per-mesh status words) and returns;
 * sh_collect waits for the oldest submitted run and reports its status like sh_run.  At most two runs may be in flight, so
 * the device goes from one run to the next without waiting for the host.  sh_run = sh_submit + sh_collect. */
int  sh_submit(sh_ctx*, uint32_t stage_mask, sh_landmarks* out /* B, host, nullable */);
int  sh_collect(sh_ctx*);
/* Device address of the B result structs of the last sh_run (for a collective gather). */
int  sh_landmarks_device(sh_ctx*, void** dev_ptr, size_t* nbytes);
/* ---- several GPUs from ONE host process (a host in C, C++ or any FFI that does not launch one process per GPU) ----
 * One sh_ctx per device; the humeri of a cohort are sharded over the contexts (independent units of work: the reference runs one
 * `Humerus(stl)` at a time, bone.py:110-131), each context runs its shard (sh_submit on every context, then sh_collect), and
 * the records come together at ctxs[0].  What crosses the xGMI links: the parameter block once (the reference loads the same
 * pickled forest / ONNX blob in every process: bicipital_groove.py:21-25, anatomic_neck.py:56-60) and the records once per step
 * -- no exchange inside the path.  RCCL is loaded at run time by sh_comm_init_all (librccl.so.1, or the path in SHOULDER_RCCL_LIB);
 * nothing else in the library needs it.  Errors of these three calls are left on ctxs[0] (sh_last_error).  The multi-process
 * launch (bench.py --gpus N, one rank per GPU under torch.distributed) does the same transfers in shoulder_amd/dist.py.
 *   sh_comm_init_all     ncclCommInitAll over the contexts' devices (distinct devices); rank i = ctxs[i].  A context leaves its group
 *                        when it is destroyed or put into another.
 *   sh_bcast_weights     the parameter block of ctxs[root] (layout of sh_param_block) to every context; each has loaded a network
 *                        and a forest of the same shape before (any values); receivers validate the block like sh_param_block_commit.
 *   sh_gather_landmarks  the records of every context's last run, rank order, to host memory at ctxs[0]: sum of the batch sizes
 *                        records of sh_record_bytes(rows) bytes each, `rows` = the record format all contexts are set to
 *                        (sh_set_record_rows; 0 = full sh_landmarks).
 * STATUS: EXPERIMENTAL for n > 1.  The build pool has one GPU per box: these calls have run on hardware as a group of ONE only
 * (tests/test_gpu_comm.py prints the world size it ran with); the n > 1 legs -- the multi-rank ncclBroadcast, the grouped
 * ncclSend / ncclRecv of the gather, destroying one rank's communicator while its siblings live -- are untested on hardware. */
int  sh_comm_init_all(sh_ctx** ctxs, int n);
int  sh_bcast_weights(sh_ctx** ctxs, int n, int root);
int  sh_gather_landmarks(sh_ctx** ctxs, int n, sh_landmarks* out_root /* host */);
/* Page-locked host memory for the `out` array of sh_run when it is reused from run to run (the reference returns fresh
 * NumPy arrays from every accessor; a streaming caller keeps one record buffer): direct D2H, no page faults. */
int  sh_host_alloc(sh_ctx*, size_t nbytes, void** out);
int  sh_host_free(sh_ctx*, void* p);

/* utils.transform_pts (utils.py:172-188) for B point sets on the device:
 * out[off[b]..off[b+1]) = T[b] * in[...]; in/out are DEVICE pointers to float64 xyz. */
int  sh_affine_apply(sh_ctx*, const double* T /* host, B x 16 */, const void* dev_in, void* dev_out,
                     const int64_t* off /* host, B+1 */, int B);
/* Trimesh.apply_transform of mesh b (bone.py:155): transformed float64 vertices -> host. */
int  sh_mesh_transformed(sh_ctx*, int b, const double* T /* 16 */, double* out_verts /* V x 3 */);
/* utils.transform_pts for one host point set (every Landmark.transform_landmark, e.g. canal.py:84,
 * anatomic_neck.py:120): upload n xyz float64, transform on the device, download. */
int  sh_transform_points(sh_ctx*, const double* T /* 16 */, const double* in_pts /* host, n x 3 */, int n, double* out_pts /* host */);

/* The closed largest loop of plane k of slice set `set` ("distal", "prox", "neckc") of humerus b after a run -- what
 * `Slices.slices[k].polygons_closed[...]` exterior holds in the reference (slice.py:53-59, surgical_neck.py:37-54): n + 1
 * points (x, y) in the box frame, counter-clockwise, canonical start (rule B-1), first = last.  Read from the fixed slot range
 * or, for a plane with more crossing segments than slots, from the overflow pool.  out == NULL or cap < n + 1: only
 * *n_out = n + 1 is set. */
int  sh_ring(sh_ctx*, const char* set, int b, int k, double* out, int cap, int* n_out);
/* `mesh_ct.section(plane_origin, plane_normal).vertices` for mesh b (AnatomicNeck.plane_points,
 * anatomic_neck.py:155-172): unique crossing points of one general plane, CT coordinates, unordered. */
int  sh_section_plane(sh_ctx*, int b, const double* origin /* 3 */, const double* normal /* 3 */, double* out_pts /* cap x 3 */,
                      int cap, int* n_out);

/* `mesh.slice_plane(origin, normal)` (HumeralHeadOsteotomy.resect_mesh, arthroplasty.py:80-87; trimesh
 * slice_faces_plane + the constructor's 8-decimal vertex merge) for ONE mesh given as host arrays in any coordinate
 * system and P planes at once (a sweep of resection planes is one call): per plane the part on the side the normal
 * points to -- kept faces, cut faces re-triangulated, unreferenced vertices dropped, vertices merged.  Vertex order is
 * this library's canonical one (by first use; trimesh's depends on its hash sort), faces keep trimesh's order.
 * out_edges (nullable): per cut face the edge between its two new vertices = the section polyline of the plane
 * (HumeralHeadOsteotomy.points, arthroplasty.py:69-78).  counts: P x (n_verts, n_faces, n_edges).
 * out_verts == NULL: count only; counts then hold capacities that suffice (n_verts is an upper bound). */
int  sh_slice_mesh_planes(sh_ctx*, const double* verts /* nv x 3 */, int nv, const int32_t* faces /* nf x 3 */, int nf,
                          const double* origins /* P x 3 */, const double* normals /* P x 3 */, int P,
                          double* out_verts /* P x cap_v x 3 */, int cap_v, int32_t* out_faces /* P x cap_f x 3 */, int cap_f,
                          int32_t* out_edges /* P x cap_e x 2 */, int cap_e, int32_t* counts /* P x 3 */);

/* ---- stage-level access for parity tests: named intermediate device buffers ----------
 * names: "verts_obb" "obb_transform" "full.zs" "full.centroids" "full.areas" "full.nloops"
 * "distal.*" "prox.*" "prox.ixy" "prox.itr_start" "prox.itr_centered_start" "canal.points"
 * "groove.X" "groove.nX" "groove.proba" "anp.image" "anp.logits" "anp.roll" ... (see sh_buffer_info) */
int  sh_buffer_info(sh_ctx*, const char* name, size_t* nbytes, int* elem_size);
/* Device address of a named buffer (e.g. "verts_obb", "verts_csys": inputs / outputs of sh_affine_apply; valid until the
 * next upload). */
int  sh_buffer_device(sh_ctx*, const char* name, void** dev_ptr, size_t* nbytes);
int  sh_fetch(sh_ctx*, const char* name, void* host, size_t nbytes);
int  sh_store(sh_ctx*, const char* name, const void* host, size_t nbytes);

/* The anatomic-neck network alone (replaces the `onnxruntime.InferenceSession.run` call of
 * humerus/anatomic_neck.py:67-76): n images [n][H][W] float32 (host) -> logits [n][H][W] float32
 * (host), computed in sh_params.unet_dtype.  H and W must be multiples of 16 << depth. */
int  sh_unet_infer(sh_ctx*, const float* images, int n, int H, int W, float* logits);

/* Average duration (ms) of the named kernel over the launches since the last reset, measured
 * with HIP events on the ctx stream (bench.py roofline); name NULL resets all timers. */
int  sh_kernel_time_ms(sh_ctx*, const char* kernel, double* avg_ms, int* launches);
/* level 0: off; 1: events around every launch; 2: around the UNet layers only (events around all ~150 launches of a
 * run stretch it by ~4 % at B = 64, so a measurement inside a timed region uses level 2). */
int  sh_enable_timing(sh_ctx*, int level);

/* Streaming use (one sh_run after another on the resident batch): with overlap on, sh_run starts a
 * background thread, once all its device work is enqueued, that computes the convex hulls the NEXT
 * sh_run(SH_STAGE_OBB) needs (the host part of mesh.py:82 `apply_obb`), so the host hulls of run
 * k+1 overlap the device work of run k.  Prepared hulls are used only if the batch is unchanged
 * (any sh_upload_meshes / sh_synth_batch / sh_store("verts") voids them); sh_discard_prepared
 * drops them explicitly (the next run then does its host phase inline).  Results are identical
 * either way.  Off by default. */
/* Where the convex hull of SH_STAGE_OBB (`Trimesh.apply_obb()` -> qhull, mesh.py:82) is computed: "host" (quickhull on a
 * process-wide pool of worker threads, points read back through the device prefilter; hidden behind the previous run with
 * sh_set_overlap), "device" (round-based quickhull, k_hull.h: nothing leaves the GPU; a humerus it gives up -- more than 8 192
 * prefilter survivors, a horizon pinched by nearly coplanar points -- is re-done ALONE from inside sh_run / sh_collect: host
 * quickhull for that humerus, its stages re-run as a window of one behind whatever else is in flight; it stays on the host
 * hull while the batch is resident) or "auto" (default, also SHOULDER_HULL: host while the rank has enough usable hardware
 * threads -- 16 for a single rank, 48 per rank with LOCAL_WORLD_SIZE > 1; affinity mask and cgroup CPU quota counted).  Both give the
 * same triangles and the same record bits, hence bit-identical frames.  sh_get_hull_mode: 0 host, 1 device. */
int  sh_set_hull_mode(sh_ctx*, const char* mode);
int  sh_get_hull_mode(const sh_ctx*);
/* What "auto" resolves to for a context created by THIS process now (0 host, 1 device): usable hardware threads (affinity mask, cgroup
 * CPU quota) divided by LOCAL_WORLD_SIZE against 16 (a rank alone on its host) / 48 (several ranks).  Needs no device: a launcher can
 * size its lanes before it creates a context (bench.py: three lanes with the device hull, two with the host hull). */
int  sh_auto_hull_mode(void);
int  sh_set_overlap(sh_ctx*, int on);
int  sh_discard_prepared(sh_ctx*);

/* Several contexts on ONE device ("lanes": own stream, own scratch; consecutive batches go to alternating contexts
 * through sh_submit / sh_collect).  Their streams overlap on the device: the launch- and latency-bound geometry
 * kernels of one run execute beside the chip-filling UNet kernels of another (measured at B = 64: 14.2 -> 11.5 ms per
 * batch with two contexts).  Two UNet passes side by side gain nothing, each just takes twice as long; contexts that
 * turn this on chain their UNet passes with events in the order the host enqueued them, so a UNet pass only ever
 * shares the device with geometry.  Results are identical either way.  Off by default.
 * The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on
 * one queue run in order: a process that also runs torch / RCCL streams should export GPU_MAX_HW_QUEUES=16 before HIP
 * initialises (bench.py does). */
int  sh_set_unet_turns(sh_ctx*, int on);

#ifdef __cplusplus
}
#endif
#endif /* SHOULDER_HIP_H */
