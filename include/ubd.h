/* ubd.h -- C ABI of libubd_hip.so: the MI355X (gfx950) implementation of the
 * ubdvss hot path (dilated-FCN forward -> threshold map -> external components ->
 * rotated quads; train step = forward + loss + backward + Adam).
 *
 * The reference (asmekal/ubdvss) has no FFI: the path sits behind three Python
 * seams.  Each entry point below names the seam it replaces (paths relative to
 * the reference root).  All pointers are raw DEVICE pointers owned by the caller
 * (e.g. torch tensors' data_ptr()) unless marked HOST; the library owns only its
 * handle.  Every call enqueues work on the caller's HIP stream (`stream` is a
 * hipStream_t passed as void*, NULL = default stream) and returns without
 * synchronising.  Return value: 0 = ok, non-zero = error (see ubd_last_error()).
 * No torch types, no C++ types.
 */
#ifndef UBD_H
#define UBD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UBD_ABI_VERSION 3

/* activation storage / compute dtype (weights, logits, loss and gradients are fp32) */
enum { UBD_F32 = 0, UBD_BF16 = 1, UBD_F16 = 2 };
/* input image dtype for ubd_forward; UBD_IN_PREPACKED may be OR-ed in: the packed-fragment region of `workspace`
 * already holds the fragments of the CURRENT parameter values (left there by a previous ubd_forward or
 * ubd_pack_weights on the same handle and workspace), so the per-call repacking kernels (~14 us) are skipped -- and so is the
 * zeroing of the fused stem kernel's strip counters, which that kernel leaves at zero itself: pass the flag only while
 * nothing else has written to the head of the workspace since. */
enum { UBD_IN_F32 = 0, UBD_IN_U8 = 1, UBD_IN_PREPACKED = 0x100 };
/* preprocessing fused into the first layer's load (net.py:217-218, NetConfig.get_preprocessing_fn) */
enum { UBD_PRE_NONE = 0, UBD_PRE_MOBILENET = 1 };

/* Mirrors the fields of NetConfig that reach the kernels (net.py:98-132):
 * grey -> c_in, fml_compatible, n_classes (0 = detection only). */
typedef struct ubd_config {
    int32_t c_in;            /* 1 (grey=True, the reference default) or 3 */
    int32_t n_classes;       /* 0..UBD_MAX_CLASSES */
    int32_t fml_compatible;  /* 1: ZeroPadding2D((1,0),(1,0)) + 'valid' for stride 2 (net.py:229-232) */
    int32_t dtype;           /* UBD_F32 / UBD_BF16 / UBD_F16 */
} ubd_config;

#define UBD_MAX_CLASSES 31
#define UBD_LOSS_FLOATS 16
#define UBD_N_FILTERS 24

typedef struct ubd_handle ubd_handle;

/* --- lifecycle ----------------------------------------------------------- */
int ubd_abi_version(void);
const char *ubd_build_id(void);                         /* 16 hex digits: fingerprint of the kernel sources (every .hip and .h file of csrc) THIS library was compiled from */
const char *ubd_last_error(void);                       /* thread-local message of the last failure */
/* Host staging helper of the streaming seam (ModelRunner.predict_stream; the loop of model_runner.py:60-67): copies n bytes with `threads`
 * native threads (1..16) -- a batch of pageable numpy memory into a pinned staging buffer without the interpreter in the way.  Returns 0. */
int ubd_host_memcpy_mt(void *dst, const void *src, size_t n, int threads);
int ubd_create(const ubd_config *cfg, ubd_handle **out); /* replaces NetManager.build_model (net.py:273-314) */
void ubd_destroy(ubd_handle *h);

/* Number of fp32 parameters; flat order = Keras model.get_weights() order
 * (SURVEY.md 9.2): per separable layer [depthwise(3,3,C,1), pointwise(1,1,C,24), bias],
 * per Conv2D [kernel HWIO, bias]; head last. */
size_t ubd_param_count(const ubd_handle *h);
/* Compute units the handle sizes its persistent grids for: the device's count, or UBD_TEST_NUM_CUS when that is set at
 * ubd_create (tests walk many tiles per block on small shapes with it; no reference counterpart). */
int ubd_num_cus(const ubd_handle *h);

/* Bytes of caller-provided device workspace needed by ubd_forward / ubd_train_step
 * for a batch of n images of height x width (multiples of 4). */
size_t ubd_forward_workspace_bytes(const ubd_handle *h, int n, int height, int width);
size_t ubd_train_workspace_bytes(const ubd_handle *h, int n, int height, int width);
size_t ubd_postprocess_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w, int cap);
size_t ubd_loss_workspace_bytes(const ubd_handle *h, int n, int map_h, int map_w);

/* --- inference ----------------------------------------------------------- */
/* Replaces keras Model.predict(images) at model_runner.py:119 / predict.py:74-76.
 * images: NHWC, (n, height, width, c_in), in_dtype UBD_IN_F32 (already preprocessed, or raw with
 *         preprocessing != NONE) or UBD_IN_U8.
 * logits: fp32 NHWC (n, height/4, width/4, 1+n_classes): channel 0 detection logit. */
int ubd_forward(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                int n, int height, int width, float *logits,
                void *workspace, size_t workspace_bytes, void *stream);

/* Layer-level entry points (profiling / roofline measurement of the dominant kernel; the
 * reference has no counterpart -- Keras runs the whole graph in one session.run).
 * ubd_pack_weights fills the packed-fragment region at the start of a forward workspace;
 * ubd_dilated_layer then runs ONE dense dilated 3x3 layer (layer = 0..5 -> net.py:298-304,
 * dilation 1,2,4,8,16,1) + bias + ReLU on (n, map_h, map_w, 24) activations. */
int ubd_pack_weights(ubd_handle *h, const float *params, void *workspace, size_t workspace_bytes, void *stream);
int ubd_dilated_layer(ubd_handle *h, const float *params, int layer, const void *in, void *out,
                      int n, int map_h, int map_w, const void *workspace, void *stream);

/* Replaces ModelRunner.predict's host tail (model_runner.py:121-134) and
 * SegmapManager.postprocess (segmap_manager.py:41-69) + utils.get_contours_and_boxes
 * (utils.py:51-60) for the whole batch:
 *   map = logits[...,0] > logit_threshold (strict);
 *   external 8-connected components (cv2.findContours RETR_EXTERNAL semantics);
 *   keep contourArea > min_area; minAreaRect -> boxPoints -> round(x*scale) half-to-even;
 *   optional class vote: argmax of mean softmax(class logits) over the filled contour.
 * Outputs (device):
 *   binary_map : int32 (n, map_h, map_w) values {0,1}              (may be NULL)
 *   quads      : int32 (n, cap, 8)  x1,y1,...,x4,y4 in cv2.boxPoints order, objects in cv2's
 *                return order (last raster-discovered first)
 *   classes    : int32 (n, cap)     (ignored when n_classes == 0; may be NULL then)
 *   counts     : int32 (n)          number of objects found; if counts[i] > cap the list of
 *                image i was truncated to cap entries and the caller must treat it as an error. */
int ubd_postprocess(ubd_handle *h, const float *logits, int n, int map_h, int map_w,
                    float logit_threshold, int scale, float min_area,
                    int32_t *binary_map, int32_t *quads, int32_t *classes, int32_t *counts, int cap,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ubd_forward of one batch and ubd_postprocess of an EARLIER batch's logits in one call (arguments as in those two; pp_* name the
 * earlier batch).  Replaces nothing new in the reference -- ModelRunner.predict (model_runner.py:105-138) runs the model and the
 * postprocess one after the other -- it is how the MI355X host overlaps the postprocess of batch k with the forward pass of
 * batch k+1: when the forward pass takes the one-kernel stem, the first pp_n blocks of that kernel do the postprocess before they
 * join the stem's work queue (no second stream, no events, no extra launch); otherwise the two calls are made back to back.
 * pp_logits must not alias `logits`.  Results are identical to the separate calls. */
int ubd_forward_postprocess(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                            int n, int height, int width, float *logits, void *workspace, size_t workspace_bytes,
                            const float *pp_logits, int pp_n, int pp_map_h, int pp_map_w, float logit_threshold, int scale,
                            float min_area, int32_t *binary_map, int32_t *quads, int32_t *classes, int32_t *counts, int cap,
                            void *pp_workspace, size_t pp_workspace_bytes, void *stream);

/* --- training ------------------------------------------------------------ */
/* Replaces the loss callable losses.get_loss(classification_mode)(y_true, y_pred)
 * (losses.py:20-24, :33-126) together with its autodiff gradient.
 *   y_true : int32 (n, map_h, map_w) labels 0..n_classes (0 = background)
 *   loss   : fp32 [UBD_LOSS_FLOATS] = {total, detection, classification, n_hard_k,
 *            positive_loss, negative_loss, hard_negative_loss (losses.py:138-191), n_pos,
 *            tp, tn, fp, fn of the detection map logit0 > 0 vs y_true > 0, number of positive pixels whose
 *            class argmax equals the label (keras_metrics.py:110-172), n_pixels, 0, 0}
 *   dlogits: fp32 like logits (may be NULL for loss only)
 * Batch-global reductions and top-k run over the n images given (per-replica semantics). */
int ubd_loss(ubd_handle *h, const float *logits, const int32_t *y_true, int n, int map_h, int map_w,
             float *loss, float *dlogits, void *workspace, size_t workspace_bytes, void *stream);

/* One pass fwd -> loss -> backward (replaces the body of Keras train_on_batch driven by
 * fit_generator, train.py:176-188, up to but excluding the optimiser update).
 *   grads : fp32 [param_count], same flat order as params (overwritten)
 *   loss  : fp32 [UBD_LOSS_FLOATS] as in ubd_loss
 * The caller may all-reduce `grads` across ranks before ubd_adam_step. */
int ubd_train_step(ubd_handle *h, const float *params, const void *images, int in_dtype, int preprocessing,
                   const int32_t *y_true, int n, int height, int width,
                   float *grads, float *loss, void *workspace, size_t workspace_bytes, void *stream);

/* Keras 2.2 Adam (train.py:110): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updated in place;
 * p -= lr_t*m/(sqrt(v)+eps); grads are multiplied by grad_scale first (1/world for DP mean). */
int ubd_adam_step(float *params, const float *grads, float *m, float *v, size_t count,
                  int t, float lr, float beta1, float beta2, float eps, float grad_scale, void *stream);

/* --- training labels --------------------------------------------------------
 * Replaces SegmapManager.build_segmentation_map (segmap_manager.py:81-104) + _proper_round (:106-133) for a whole batch:
 * every object quad (8 float64 x1,y1..x4,y4 in IMAGE pixels; fractional after _rescale_image_and_markup / augmentation) is
 * divided by `scale` in double precision, its corners are snapped outward and the polygon is filled with PIL's
 * ImageDraw.polygon rule, objects in order (later over earlier), into an int32 map of (map_h, map_w) = image size / scale.
 * values[i][o]: what to write (class id + 1, or 1; 255 for drawing).
 *   quads float64 (n, cap, 8), values int32 (n, cap), counts int32 (n) (objects per image, <= cap), labels int32 (n, map_h, map_w):
 * the y_true layout of ubd_loss / ubd_train_step.  Bit-identical to Pillow for every quad except zero-area folds whose
 * opposite corners coincide (see raster.hip); the Python host rejects those. */
int ubd_build_label_maps(const double *quads, const int32_t *values, const int32_t *counts, int n, int cap,
                         int map_h, int map_w, int scale, int32_t *labels, void *stream);

/* --- data parallelism (no reference counterpart: the reference is single-device) -------------------
 * One process per GPU, per-replica loss (losses.py:86-126 applied to the rank's own images), ONE sum all-reduce of the flat
 * fp32 gradient vector per step over RCCL / xGMI, 1/world applied by ubd_adam_step's grad_scale, parameters broadcast once.
 * The handle owns the communicator.  unique_id: 128 HOST bytes created on one rank by ubd_comm_unique_id and handed to all
 * ranks by the caller (file, socket, launcher store).  librccl is resolved with dlopen when the first of these is called. */
#define UBD_UNIQUE_ID_BYTES 128
/* flags of ubd_comm_init.  UBD_COMM_FUSED: ubd_train_step all-reduces `grads` itself (SUM over ranks) -- the dilated + head
 * segment on a communication stream under the stem layers' backward pass, the stem segment on the caller's stream, which
 * then joins the first -- so `grads` come back summed and the caller must NOT call ubd_allreduce_grads again. */
enum { UBD_COMM_FUSED = 1, UBD_COMM_GLOBAL_LOSS = 2 };
/* UBD_COMM_GLOBAL_LOSS: ubd_loss / ubd_train_step evaluate the reductions of losses.py:86-126 -- n_pos, n_neg, the positive /
 * negative means, the top-k of the FLATTENED batch (losses.py:111) and the classification mean -- over the images of all
 * ranks (rank r holds flat indices [r*npix, (r+1)*npix), equal shards), i.e. the reference's loss at the global batch
 * instead of one loss per replica.  The returned loss is the global one on every rank and d loss / d logits is its exact
 * gradient, so the parameter gradients must be SUMMED over the ranks and applied with grad_scale = 1 (not 1/world). */
int ubd_comm_unique_id(void *unique_id_out);
int ubd_comm_init(ubd_handle *h, const void *unique_id, int rank, int world, int flags);   /* collective: every rank calls it */
int ubd_comm_destroy(ubd_handle *h);                                                        /* also done by ubd_destroy */
int ubd_comm_world(const ubd_handle *h);                                                    /* 1 without a communicator */
/* In-place SUM all-reduce of grads[0..count) over the ranks, enqueued on `stream` (replaces nothing in the reference; it is
 * the exchange step between ubd_train_step and ubd_adam_step). */
int ubd_allreduce_grads(ubd_handle *h, float *grads, size_t count, void *stream);
/* params[0..count) of rank `root` to every rank (initial weights / after loading a model on one rank). */
int ubd_broadcast_params(ubd_handle *h, float *params, size_t count, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UBD_H */
