/*
 * fdt.h -- C ABI of libfdt_hip.so: the MI355X (gfx950) implementation of the per-frame
 * face-detection-and-tracking hot path of limacv/Face-detection-and-tracking.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; these entry points are
 * what a binding for each reference function on the path attaches to.
 * Each declaration cites the reference interface it replaces (file:line under the
 * reference root).  Plain pointers and sizes only: no torch / numpy types cross this
 * boundary.  INTEGRATION.md shows the ctypes stubs the reference side would add.
 *
 * Conventions
 *   - every function returns FDT_OK (0) or a negative FDT_ERR_* code; the message for the
 *     calling thread's last failure is fdt_last_error().
 *   - "host" entry points take host pointers, copy in/out and synchronise before returning.
 *     "_dev" entry points take device pointers plus a hipStream_t (as void*) and only enqueue work.  NULL = the
 *     handle's own stream where the call has a handle (fdt_model_*, fdt_tracker_*), else the calling thread's
 *     private stream (fdt_thread_stream); never the legacy stream.
 *   - caller owns every in/out buffer; the library owns weights and workspaces.
 *   - a handle is not thread-safe; distinct handles (fdt_model_clone handles included) may be used from distinct
 *     threads at the same time, and so may the host entry points.  The library never uses the legacy (null) stream:
 *     host entry points run on a private stream of the calling thread, handles on their own stream.
 */
#ifndef FDT_H_
#define FDT_H_

#ifdef __cplusplus
extern "C" {
#endif

#define FDT_OK 0
#define FDT_ERR_ARG (-1)    /* bad argument (the reference raises ValueError / returns None) */
#define FDT_ERR_HIP (-2)    /* a HIP runtime call failed (no device, OOM, launch failure)   */
#define FDT_ERR_STATE (-3)  /* call order: weights missing, model not finalized, ...       */
#define FDT_ERR_NAME (-4)   /* unknown tensor name (strict load_state_dict)                */

/* arch of fdt_model_create */
#define FDT_ARCH_RES50 0    /* pyramid.py:367-374 build_sfd -> SFD(Bottleneck,[3,4,6,3])   */
#define FDT_ARCH_TRY3 1     /* pyramid_mb2_try3.py:359-366 build_sfd_mobile               */
#define FDT_ARCH_FACEBOX 2  /* FACEBOX/networks.py:60-116 FaceBox                          */
#define FDT_ARCH_TRY4 3     /* pyramid_mb2_try4.py:363-370 build_sfd_mobile (7x7 stem)     */
#define FDT_ARCH_TRY5 4     /* pyramid_mb2_try5.py:363-370 build_sfd_mobile                */
#define FDT_ARCH_TRY1 5     /* pyramid_mobile_try1.py:138-369 SFD_mobile (Mobilenetv1/v2 blocks, 6 sources) */
#define FDT_ARCH_TRY2 6     /* pyramid_mobile_try2.py:141-398 SFD_mobile (+ layerN_adj 1x1 convs)           */

/* frame formats of fdt_model_forward */
#define FDT_FRAME_U8_HWC_BGR 0 /* raw video frame; mean (104,117,123) is subtracted on device:
                                  iouTracke_cal.py:40-46 / My_test.py:23-29                  */
#define FDT_FRAME_F32_NCHW 1   /* what the reference hands to net(x): already mean-subtracted */

#define FDT_F32 0
#define FDT_F64 1

typedef struct fdt_model fdt_model;
typedef struct fdt_tracker fdt_tracker;
typedef struct fdt_comm fdt_comm;

/* ------------------------------------------------------------------ library / device */
const char* fdt_last_error(void);
int fdt_version(void);
int fdt_device_count(int* n);
int fdt_device_name(int dev, char* buf, int buflen);
int fdt_set_device(int dev);
int fdt_device_synchronize(void);
/* The calling host thread's private non-blocking stream (hipStream_t as void*): what the host entry points run on
 * and what a NULL `stream` argument of a handle-less "_dev" entry point (fdt_detect_dev, fdt_allgather_dets, the
 * consumer_stream of fdt_model_async_record) means.  A caller that passed NULL orders later work behind the
 * library's by using / synchronising this stream.                                                            */
int fdt_thread_stream(void** stream);
/* A stream confined to partition `part` of `parts` (1, 2, 4) equal shares of every XCD's compute units
 * (hipExtStreamCreateWithCUMask): forwards enqueued on streams of different partitions run side by side.
 * Destroy with fdt_stream_destroy once nothing is in flight on it.                                      */
int fdt_stream_create_partition(int part, int parts, void** stream);
int fdt_stream_destroy(void* stream);
/* free / total HBM of the current device (hipMemGetInfo) */
int fdt_device_mem_info(long long* free_bytes, long long* total_bytes);

/* ------------------------------------------------------------------ SSD post-processing ops
 * Stand-alone, no model handle: these are what `from layers import *` /
 * `from layers.box_utils import decode, nms` resolve to.                                   */

/* PriorBoxLayer.__call__(prior_idx, f_width, f_height)  layers/functions/prior_box.py:28-44.
 * One level: for i<f_h, j<f_w, scale<n_scales: (cx,cy,w,h) then one more box per aspect
 * ratio; computed in f64 and rounded once to f32 like torch.Tensor(list).
 * out: [f_h*f_w*n_scales*(1+n_ar), 4] f32 host.                                             */
int fdt_priorbox(int width, int height, int stride, int box, int n_scales,
                 const double* aspect_ratios, int n_ar, int f_w, int f_h, float* out);

/* decode(loc, priors, variances)  layers/box_utils.py:238-258.  [P,4],[P,4] -> [P,4] host.  */
int fdt_decode(const float* loc, const float* priors, int P, float var0, float var1,
               float* boxes);

/* nms(boxes, scores, overlap, top_k)  layers/box_utils.py:275-340.
 * keep: [n] int64 zero-padded (like the reference's `keep` tensor), *count = #kept.          */
int fdt_nms(const float* boxes, const float* scores, int n, float overlap, int top_k,
            long long* keep, int* count);

/* Detect(num_classes,bkg,top_k,conf_thresh,nms_thresh)(loc,conf,priors)
 * layers/functions/detection.py:15-84.  loc [B,P,4], conf [B,P,C] (already softmaxed),
 * priors [P,4]; out [B,C,top_k,5] rows (score,x1,y1,x2,y2), zero padded; counts [B,C]
 * (may be NULL).  nms_thresh <= 0 -> FDT_ERR_ARG (the reference raises ValueError, :28-29).  */
int fdt_detect(const float* loc, const float* conf, const float* priors, int B, int P,
               int num_classes, int top_k, float conf_thresh, float nms_thresh, int nms_top_k,
               float var0, float var1, float* out, int* counts);
int fdt_detect_dev(const float* loc, const float* conf, const float* priors, int B, int P,
                   int num_classes, int top_k, float conf_thresh, float nms_thresh,
                   int nms_top_k, float var0, float var1, float* out, int* counts,
                   void* workspace, long long workspace_bytes, void* stream);
long long fdt_detect_workspace_bytes(int B, int P, int nms_top_k);

/* calculate_iou(box_a, box_b)  utils/calc_performance.py:54-74 (+ intersect :4-31).
 * a [A,4], b [B,4] (x1,y1,x2,y2) -> out [A,B]; dtype FDT_F32 / FDT_F64 follows the input
 * like numpy does (the tracker feeds f64).  0/0 -> NaN, no epsilon.                          */
int fdt_pairwise_iou(const void* a, int A, const void* b, int B, int dtype, void* out);
/* calculate_distance(box_a, box_b)  utils/calc_performance.py:34-51: the association measure of the tracker's
 * `use_iou = False` branch (iouTracke_cal.py:136-138).  Same shapes / dtypes as fdt_pairwise_iou.  The final
 * `dis ** 0.25` is correctly rounded here; numpy's pow is only faithful (1 ulp, libm dependent), so parity with a
 * reference run is stated as 1 ulp.                                                    */
int fdt_pairwise_distance(const void* a, int A, const void* b, int B, int dtype, void* out);

/* FaceBoxes anchors: DataEncoder.__init__  FACEBOX/encoderl.py:12-48.  out [21824,4] (cx,cy,w,h)/1024. */
int fdt_facebox_anchors(float* out);
/* DataEncoder.decode_np(loc, conf, conf_thres) + nms_np(threshold)  FACEBOX/encoderl.py:308-325,217-266.
 * loc [P,4], conf [P,2] (softmaxed), anchors [P,4] -> boxes [<=P,4] (x1,y1,x2,y2 in [0,1], keep
 * order) and probs; *count = number kept.  No candidate cap, no output cap (like the reference).      */
int fdt_facebox_decode(const float* loc, const float* conf, const float* anchors, int P,
                       float conf_thresh, float nms_thresh, float* boxes, float* probs, int* count);

/* nn.Conv2d / F.conv2d as the nets use it (pyramid.py:14,35-39,58-59,83-94; pyramid_mb2_try3.py:16,98,
 * 113; FACEBOX/networks.py:13,63-66), NCHW f32, OIHW weights, with the fusions of the forward graphs:
 * + bias, + residual [B,Cout,Ho,Wo], + bilinear x2 (align_corners=False) upsample of up [B,Cout,uh,uw]
 * (pyramid.py:65-68), then act (0 none, 1 ReLU, 2 ReLU6).  (ksize,stride,pad,dil) must be one of the
 * instantiated classes: 1x1 s1|s2 p0; 3x3 s1 p1; 3x3 s1 p2 d2; 3x3 s2 p1; 7x7 s2 p3; 7x7 s4 p3; 5x5 s2 p2.
 * tile/ksplit pick a kernel variant explicitly (tile < 0, ksplit <= 0: automatic; tile >= 100 means
 * variant_class * 100 + tile for the alternative implementations of a class, e.g. deep-stage 1x1;
 * ksplit | FDT_SPLIT_COMBINE: the split-K partial sums are combined inside the conv kernel by the
 * last workgroup to arrive at an output tile instead of by a second pass -- same sums, same order).    */
#define FDT_SPLIT_COMBINE 0x1000
int fdt_conv2d(const float* x, int B, int Cin, int H, int W, const float* w_oihw, const float* bias,
               int Cout, int ksize, int stride, int pad, int dil, const float* residual, const float* up,
               int up_h, int up_w, int act, int tile, int ksplit, float* out);

/* conv[0..5] of an InvertedResidual block with expand_ratio != 1 (pyramid_mb2_try3.py:96-114): 1x1 expand + BN + ReLU6,
 * depthwise 3x3 (stride 1 | 2, pad 1) + BN + ReLU6, as ONE kernel whose expanded tensor never leaves the CU.  The eval
 * BatchNorms are already folded: w1 [hid][Cin], b1 [hid], wdw [hid][9], bdw [hid].  x [B,Cin,H,W] -> out [B,hid,Ho,Wo]
 * (host pointers; the detector graphs use the same kernel on device tensors).  Cin even.                          */
int fdt_expand_dw(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                  const float* bdw, int hid, int stride, float* out);
/* The whole InvertedResidual (pyramid_mb2_try3.py:96-134, expand_ratio != 1, BatchNorms folded) in one launch: expand, depthwise,
 * 1x1 project (wp [oup][hid], bp [oup]; oup <= 32) and, residual != 0 (stride 1, oup == Cin), + x.  out: [B][oup][Ho][Wo].      */
int fdt_ir_block(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                 const float* bdw, int hid, int stride, const float* wp, const float* bp, int oup, int residual, float* out);

/* ------------------------------------------------------------------ IoU tracker
 * The inline tracker of iouTracke_cal.py:113-156 (per frame) and :174-177 (finalise), as a
 * device-resident state machine.  A track is {bboxes, max_score, start_frame}.  One frame's detections
 * and association state live in LDS, which bounds max_dets at 1512 (the reference emits <= 2*750).  */
fdt_tracker* fdt_tracker_create(double sigma_iou, double sigma_h, int t_min, int max_dets,
                                int log_frames);
void fdt_tracker_destroy(fdt_tracker* t);
int fdt_tracker_reset(fdt_tracker* t);
/* one frame of detections [n,5] f64 host rows (x1,y1,x2,y2,score) == `det0` at :124          */
int fdt_tracker_step(fdt_tracker* t, const double* dets, int n);
/* same, but reads the device-resident Detect output [C,top_k,5] of frame `image` directly and
 * performs the host unpack of iouTracke_cal.py:53-84 on device (rows while score >= thr,
 * boxes * (w,h,w,h) in f32, dummy row if none).  Asynchronous on `stream`.                   */
int fdt_tracker_step_dev(fdt_tracker* t, const float* det_out, int num_classes, int top_k,
                         int width, int height, float score_thresh, void* stream);
/* The n_frames frames of one frame-parallel step (rank order == frame order after fdt_allgather_dets) associated
 * in ONE launch; frame g's Detect record is at det_out + g*stride_floats.  Bit-identical to n_frames calls of
 * fdt_tracker_step_dev, i.e. to n_frames iterations of the loop at iouTracke_cal.py:117-156.  n_frames <= log_frames.
 * Steps issued on different streams are ordered by the tracker itself (it is one sequential state machine).        */
int fdt_tracker_step_dev_multi(fdt_tracker* t, const float* det_out, int n_frames, long long stride_floats,
                               int num_classes, int top_k, int width, int height, float score_thresh,
                               void* stream);
/* iouTracke_cal.py:174-175, then copy the event log back.  After this the accessors work.    */
int fdt_tracker_finish(fdt_tracker* t);
int fdt_tracker_num_tracks(fdt_tracker* t, int* n);
int fdt_tracker_track_info(fdt_tracker* t, int idx, int* n_boxes, double* max_score,
                           int* start_frame);
int fdt_tracker_track_boxes(fdt_tracker* t, int idx, double* boxes /* [n_boxes,4] */);
/* Which of the kernel's two association forms the frames since create / reset ran (both make the decisions of the loop at
 * iouTracke_cal.py:129-148 bit for bit; this makes the switch observable): *frames = frames stepped; form_frames[0] = candidate
 * form (<= 6 detections above sigma_iou per track), [1] = exact form because some pair's IoU was NaN (0/0: a zero-area box against
 * a zero-area box, :73-74's dummy row), [2] = exact form because a track had more than 6 candidates, [3] = exact form because
 * sigma_iou < 0.  Waits for the steps enqueued so far.  Either output may be NULL.                                          */
int fdt_tracker_stats(fdt_tracker* t, long long* frames, long long* form_frames /* [4] */);

/* ------------------------------------------------------------------ detector model
 * build_sfd('test',640,2) / build_sfd_mobile('test',640,2) / FaceBox()                       */
fdt_model* fdt_model_create(int arch, int device);
void fdt_model_destroy(fdt_model* m);
/* A second handle on the same net and GPU that shares the weights of `src` (state dict, BN-folded and tiled device
 * copies) but owns its stream, activations, kernel plan and Detect workspace: one per frame in flight costs one weight
 * copy per GPU instead of one per handle.  `src` must be finalized; detect / priorbox settings and plan hints are
 * copied.  While clones exist fdt_model_set_tensor fails with FDT_ERR_STATE on all of them (weights are read-only).
 * The reference has one module object per process (iouTracke_cal.py:94-107); this is its multi-stream equivalent.   */
fdt_model* fdt_model_clone(fdt_model* src);
/* net.load_state_dict(d): one call per (key, tensor); unknown key -> FDT_ERR_NAME.
 * data is f32 (num_batches_tracked may be passed with ndim 0 and is ignored).               */
int fdt_model_set_tensor(fdt_model* m, const char* name, const float* data, int ndim,
                         const long long* dims);
/* number of state-dict keys still missing (strict load) and the i-th of them                 */
int fdt_model_missing(fdt_model* m, int* n);
int fdt_model_missing_name(fdt_model* m, int i, char* buf, int buflen);
/* fold BN (eval, eps 1e-5), re-tile weights for the conv kernels, upload.                    */
int fdt_model_finalize(fdt_model* m);
/* net.priorbox = PriorBoxLayer(width,height,stride,box) + net.firstTime = True
 * (iouTracke_cal.py:98,103; My_test.py:31-35).  n_levels strides/boxes.                      */
int fdt_model_set_priorbox(fdt_model* m, int width, int height, int n_levels,
                           const int* stride, const int* box);
/* net.detect = Detect(2,0,top_k,conf_thresh,nms_thresh)  (My_test.py:36)                     */
int fdt_model_set_detect(fdt_model* m, int top_k, float conf_thresh, float nms_thresh,
                         int nms_top_k);
/* Fused ingest (on: 0 off, 1 = default: the stride-2 stem of Res50, 2 = also the stride-4 stem of FaceBoxes, where it measured
 * slower and is therefore not the default; env FDT_FUSE_INGEST at create time): with uint8 frames and a 7x7 stem the
 * (float)u8 - mean (/ 255) of iouTracke_cal.py:40-46 / My_test_facebox.py:14-15 happens inside the stem convolution's staging
 * (csrc/conv_stem_u8.h) -- no f32 NCHW copy of the frame is written.  Same bits as the two-launch form (on = 0).  Tensor
 * "input" is then formed on request (fdt_model_get_tensor) from the uint8 frames of the last forward.                     */
int fdt_model_fuse_ingest(fdt_model* m, int on);
int fdt_model_get_detect(fdt_model* m, int* top_k, float* conf_thresh, float* nms_thresh, int* nms_top_k);   /* any may be NULL */
/* y = net(x)  pyramid.py:218-351 (Res50), pyramid_mb2_try3.py:218-340 (try3).
 * frames: B images in `format`; out: [B,2,top_k,5] f32; counts: [B,2] or NULL.               */
int fdt_model_forward(fdt_model* m, const void* frames, int format, int B, int H, int W,
                      float* out, int* counts);
int fdt_model_forward_dev(fdt_model* m, const void* frames_dev, int format, int B, int H, int W,
                          float* out_dev, int* counts_dev, void* stream);
/* Device-side frame ingest: frames are B raw u8 BGR HWC images of src_h x src_w;
 * they are resized on the GPU to W x H like cv2.resize(image, (W, H)) (INTER_LINEAR, 8-bit fixed-point
 * path; iouTracke_cal.py:123) and mean-subtracted in the same kernel, then the forward runs as in
 * fdt_model_forward.  frames_on_device != 0: device pointers for frames/out/counts, async on `stream`.
 * Agreement with cv2 itself is unpinned (cv2 is not available to the build); see DESIGN.md.        */
int fdt_model_forward_resized(fdt_model* m, const void* frames, int frames_on_device, int B, int src_h,
                              int src_w, int H, int W, float* out, int* counts, void* stream);
/* Pipelined host ingest (iouTracke_cal.py:119-124: frames reach the detector as host arrays).  forward_async copies the
 * caller's (pageable) frames into a pinned slot -- the caller may reuse its buffer at once --, enqueues the H2D copy, the
 * forward (+ the device-side resize when src_h x src_w differs from H x W; 0 = none) and the D2H of the Detect record
 * on the handle's stream and returns a ticket without waiting.  Two tickets per handle may be in flight; more frames in
 * flight = more fdt_model_clone handles (whose forwards the copy then overlaps).  fdt_model_wait blocks until the
 * ticket's record [B,2,top_k,5] / counts [B,2] are on the host (out / counts may be NULL) and frees the slot.
 * fdt_model_async_record hands the record over ON THE DEVICE: consumer_stream waits for the forward, *record_dev stays
 * valid until the ticket is retired; the consumer_stream argument of fdt_model_wait / fdt_model_release (may be NULL)
 * orders what was enqueued there before the slot's next forward.  fdt_model_release retires a ticket WITHOUT a host
 * wait and without copying anything back (device-side consumers only).                                                */
int fdt_model_forward_async(fdt_model* m, const void* frames, int format, int B, int H, int W, int src_h, int src_w,
                            int* ticket);
int fdt_model_async_record(fdt_model* m, int ticket, float** record_dev, void* consumer_stream);
int fdt_model_wait(fdt_model* m, int ticket, float* out, int* counts, void* consumer_stream);
int fdt_model_release(fdt_model* m, int ticket, void* consumer_stream);
/* Network output without Detect: loc [B,P,4], conf [B,P,2].  PyramidBox: conf is softmaxed
 * (pyramid.py:332).  FaceBox.forward (FACEBOX/networks.py:87-116): conf is the raw conf_preds.       */
int fdt_model_forward_raw(fdt_model* m, const void* frames, int format, int B, int H, int W,
                          float* loc, float* conf);
/* detect(im)  FACEBOX/My_test_facebox.py:12-36 after the resize: /255, FaceBox forward, softmax,
 * decode_np + nms_np.  frames must be 1024x1024.  boxes [B,21824,4], probs [B,21824], counts [B].
 * The _dev form leaves boxes/probs on the device (fdt_model_get_tensor "fb_boxes"/"fb_probs").       */
int fdt_model_detect_facebox(fdt_model* m, const void* frames, int format, int B, int H, int W,
                             float conf_thresh, float nms_thresh, float* boxes, float* probs, int* counts);
int fdt_model_detect_facebox_dev(fdt_model* m, const void* frames_dev, int format, int B, int H, int W,
                                 float conf_thresh, float nms_thresh, int* counts_dev, void* stream);
/* The same with line :13 of the reference's detect(), im = cv2.resize(im, (1024, 1024)), done on the GPU: frames are
 * B raw u8 BGR images of src_h x src_w (BASELINE config 5: 2160 x 3840).  frames_on_device != 0: device pointers for
 * frames / counts, asynchronous on `stream`, boxes / probs stay on the device ("fb_boxes" / "fb_probs").            */
int fdt_model_detect_facebox_resized(fdt_model* m, const void* frames, int frames_on_device, int B, int src_h,
                                     int src_w, float conf_thresh, float nms_thresh, float* boxes, float* probs,
                                     int* counts, void* stream);
/* after a forward: number of priors, and a named activation of the last forward
 * ("loc","conf","priors","c2".."c7","src0".."src5", ...) for stage-level parity tests.       */
int fdt_model_num_priors(fdt_model* m, int* P);
int fdt_model_get_tensor(fdt_model* m, const char* name, float* out, long long max_elems,
                         long long* dims4);
/* After a forward at the shape of interest: time every instantiated (tile, split-K) variant of each
 * conv layer on its real buffers (min of `iters` runs) and keep the fastest for later forwards at
 * that shape.  Results stay within the f32 tolerance but are re-associated (not bitwise) vs the
 * untuned plan.                                                                                  */
int fdt_model_autotune(fdt_model* m, int iters);
/* Persist / restore the per-layer kernel choice as text ("shape B H W" then "layer kind tile split" lines)
 * so a tuned plan is reproducible across processes.  export: *needed = bytes incl. NUL; buf may be NULL. */
int fdt_model_export_plan(fdt_model* m, char* buf, int buflen, int* needed);
int fdt_model_import_plan(fdt_model* m, const char* text);
/* After the first eager forward of a plan the launches behind the ingest kernel (all convs ... Detect) are captured
 * into a HIP graph per (output buffer, thresholds) and replayed with one hipGraphLaunch.  on = 0 returns to eager
 * launches (also: env FDT_GRAPH=0 at create time).  Per-op profiling always runs eagerly.  Results are identical.    */
int fdt_model_enable_graph(fdt_model* m, int on);
/* per-op timing of the next forwards (HIP events around every launch on the model stream).
 * fdt_model_profile_read: fills up to max entries; returns count in *n.  Entries: one per op of the plan, then
 * "detect" (decode + NMS) and "ingest" (the u8 -> f32 NCHW / resize kernel in front of the first op).            */
int fdt_model_profile_enable(fdt_model* m, int on);
int fdt_model_profile_read(fdt_model* m, int max, char* names /* max*48 */, float* ms,
                           double* flops, int* n);
/* Segment timing: ONE event pair around the contiguous ops [first_op, last_op] (indices of fdt_model_profile_read) of the
 * next forwards, no event between two kernels and the production launch sequence (lazy grouped reduce passes; the
 * segment's own deferred passes are flushed inside it): what a run of layers -- e.g. the backbone the MFMA bar of
 * BASELINE.json is stated on -- takes back to back on one stream, which the sum of per-op event intervals overstates by
 * one event packet per launch.  first_op < 0 switches profiling off.  Eager launches, like per-op profiling.         */
int fdt_model_profile_segment(fdt_model* m, int first_op, int last_op);
int fdt_model_profile_segment_ms(fdt_model* m, float* ms);
/* Algorithmic HBM bytes of one forward of the current plan (every op reads its inputs and writes its output once,
 * f32; convs read their weights once): totals, and per op in the order of fdt_model_profile_read.               */
int fdt_model_traffic(fdt_model* m, double* act_bytes, double* weight_bytes, int max, double* per_op, int* n);
/* algorithmic conv FLOPs (2*MAC, live convs only) of one frame at the last forward's size    */
int fdt_model_flops(fdt_model* m, double* flops);

/* ------------------------------------------------------------------ multi-GPU exchange (RCCL over xGMI)
 * The reference is single-GPU (its only trace of more is the commented nn.DataParallel at MyTrain_repo.py:71).  The
 * path shards by frame (frame f -> rank f mod G); the ONE exchange per step is an all-gather of each rank's fixed-size
 * Detect record [num_classes, top_k, 5] f32 (layers/functions/detection.py:48,82), after which the sequential
 * association of iouTracke_cal.py:117-156 runs over the gathered records in rank order (fdt_tracker_step_dev_multi).
 * One process per GPU: rank 0 calls fdt_comm_unique_id and ships the FDT_COMM_ID_BYTES bytes to the other ranks over
 * any host channel, then every rank calls fdt_comm_init_rank.  One process driving n GPUs: fdt_comm_init_all, and one
 * fdt_allgather_dets per local device between fdt_comm_group_begin / fdt_comm_group_end.                            */
#define FDT_COMM_ID_BYTES 128
int fdt_comm_unique_id(char* id_out /* [FDT_COMM_ID_BYTES] */);
fdt_comm* fdt_comm_init_rank(int world, int rank, const char* id /* [FDT_COMM_ID_BYTES] */, int device);
/* LOOP-BACK id: the `world` ranks are host THREADS of one process that share one GPU (RCCL refuses two ranks on a device).
 * Every thread passes the same id to fdt_comm_init_rank (which returns when all have joined) and then uses the communicator
 * exactly like an RCCL one: fdt_allgather_dets is a collective -- a host rendezvous of the ranks, then `world` device-to-device
 * copies on the rank's own stream, ordered by events; a rank missing for 120 s fails the call on all ranks.  It exists so that
 * the world > 1 branch of fdt_pipeline_* runs on a one-GPU box (tests/test_gpu_cabi_pipeline.py); it is not a transport.     */
int fdt_comm_unique_id_local(char* id_out /* [FDT_COMM_ID_BYTES] */);
fdt_comm* fdt_comm_init_all(int n_dev, const int* dev_ids);
int fdt_comm_world(fdt_comm* c, int* world, int* n_local);
int fdt_comm_group_begin(void);
int fdt_comm_group_end(void);
/* all_dev[r*floats_per_rank ...] = rank r's local_dev[0 .. floats_per_rank); device pointers, enqueued on `stream`.
 * local_index: which local device of the communicator (0 for fdt_comm_init_rank).                                   */
int fdt_allgather_dets(fdt_comm* c, int local_index, const float* local_dev, float* all_dev,
                       long long floats_per_rank, void* stream);
void fdt_comm_destroy(fdt_comm* c);

/* ------------------------------------------------------------------ the per-GPU detect + track pipeline
 * The loop of iouTracke_cal.py:117-156 as it is timed: `inflight` frames in flight, slot k = step % inflight owns a model
 * handle (m itself and fdt_model_clone()s of it: one weight copy), a HIP stream, its Detect record [batch][2][top_k][5] and
 * candidate counts; the exchange (world > 1: fdt_allgather_dets on `comm`) and the strictly sequential association run on one
 * more stream; HIP events order  detect(step i, slot k) -> exchange + track(step i) -> detect(step i + inflight, slot k).
 * Frames are DEVICE pointers to `batch` raw u8 BGR HWC frames (src_h x src_w > 0: raw source frames, resized on the GPU
 * like iouTracke_cal.py:123; else height x width); nothing waits for the host until fdt_pipeline_sync.  Same launches, same
 * bits as calling fdt_model_forward_dev + fdt_tracker_step_dev_multi one frame at a time (tests/test_gpu_cabi_pipeline.py:
 * a plain-C driver without any other GPU runtime).  m must be finalized with its PriorBox / Detect set; plan_text (may be
 * NULL) = fdt_model_import_plan for every slot.  m is borrowed and must outlive the pipeline; so must comm.            */
typedef struct fdt_pipeline fdt_pipeline;
fdt_pipeline* fdt_pipeline_create(fdt_model* m, int device, int height, int width, int inflight, int batch,
                                  const char* plan_text, fdt_comm* comm, int rank, int world, int src_h, int src_w,
                                  float score_thresh /* iouTracke_cal.py:61: 0.4 */, double sigma_iou, double sigma_h,
                                  int t_min, int log_frames);
void fdt_pipeline_destroy(fdt_pipeline* p);
/* initialisation, not a step: every slot's forward twice (plan + weight tiling, HIP-graph capture); tracker untouched */
int fdt_pipeline_prime(fdt_pipeline* p, const void* frames_dev);
/* enqueue step i: detection of `batch` frames on slot i % inflight, exchange, association of the step's world * batch frames */
int fdt_pipeline_step(fdt_pipeline* p, long long i, const void* frames_dev);
/* the same from HOST frames (iouTracke_cal.py:119-124): n_valid pageable uint8 frames are copied to a pinned landing buffer of
 * the slot (the caller's memory is free on return; the landing buffers of all slots are allocated by the first call), H2D on
 * the slot's stream in front of its forward.  Frame order is that of fdt_pipeline_step (a rank's batch = consecutive frames).
 * n_valid < batch: a partly filled last batch -- only the first n_valid frames reach the tracker; defined at world 1 only
 * (FDT_ERR_ARG otherwise: the ranks would have to know each other's n_valid).                                              */
int fdt_pipeline_step_host(fdt_pipeline* p, long long i, const void* frames_host, int n_valid);
/* frames handed over ONE at a time (what a video source delivers), executed `batch` at a time: frame i of this rank is
 * copied into the staging batch (allocated by fdt_pipeline_create) of slot (i / batch) % inflight; the batch's last frame
 * launches it (one launch per layer for `batch` consecutive frames).  fdt_pipeline_flush runs a partly filled batch (end of
 * the video) and CLOSES it.  The frames of a group arrive in order starting with i % batch == 0; a frame that would continue
 * a closed group, skip an entry or open a group while another is open is refused with FDT_ERR_STATE.  World > 1: frame i of
 * rank r is frame i * world + r of the video (the tracker sees (group, entry, rank) order), and every rank hands over and
 * flushes the same number of frames.                                                                                    */
int fdt_pipeline_step_frame(fdt_pipeline* p, long long i, const void* frame_dev);
int fdt_pipeline_flush(fdt_pipeline* p);
int fdt_pipeline_sync(fdt_pipeline* p);
/* the pipeline's device tracker: fdt_tracker_finish / _num_tracks / _track_info / _track_boxes after fdt_pipeline_sync */
fdt_tracker* fdt_pipeline_tracker(fdt_pipeline* p);
/* what slot k owns (any output may be NULL): its handle, stream (hipStream_t), record, gathered records, counts */
int fdt_pipeline_slot(fdt_pipeline* p, int slot, fdt_model** model, void** det_stream, float** record_dev,
                      float** gathered_dev, int** counts_dev);
/* timing events on the tracker stream: mark(0) ... steps ... mark(1); elapsed_ms waits for mark 1 */
int fdt_pipeline_mark(fdt_pipeline* p, int which);
int fdt_pipeline_elapsed_ms(fdt_pipeline* p, float* ms);
/* Completion stamps for latency measurements (off by default): the next n exchange + association groups -- one per
 * fdt_pipeline_step / _step_host call, one per launched group of _step_frame -- each record a timing event on the tracker
 * stream behind the association (iouTracke_cal.py:126-156 done for those frames).  stamps_read: milliseconds from mark(0) to
 * every stamp recorded so far (waits for them).  n = 0: off.                                                            */
int fdt_pipeline_stamps_enable(fdt_pipeline* p, int n);
int fdt_pipeline_stamps_read(fdt_pipeline* p, float* ms_since_mark0, int max, int* count);
/* plain device buffers (hipMalloc / hipFree / blocking copies on the calling thread's private stream) for callers that
 * link no GPU runtime of their own                                                                                  */
int fdt_dev_malloc(void** ptr, long long bytes);
int fdt_dev_free(void* ptr);
int fdt_dev_upload(void* dst_dev, const void* src_host, long long bytes);
int fdt_dev_download(void* dst_host, const void* src_dev, long long bytes);

#ifdef __cplusplus
}
#endif
#endif /* FDT_H_ */
