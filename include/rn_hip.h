/*
 * rn_hip.h -- C ABI of librn_hip.so: the MI355X (gfx950) kernels behind the RetinaNet hot path.
 *
 * The reference (vshmyhlo/retinanet-tensorflow) has no FFI seam: its model/loss code calls
 * TensorFlow kernels directly.  This library is the drop-in for exactly those kernels; every
 * entry point cites the reference call site(s) whose TF op it replaces.  The Python host side
 * (retinanet-tensorflow_amd/) binds it with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*.
 *   - all tensors are dense fp32, NHWC activations, HWIO conv kernels, [kh,kw,C] depthwise.
 *   - the caller owns every buffer (including workspaces); the library never allocates device
 *     memory, never synchronises and keeps no pointer after a call returns: every call only
 *     enqueues work on `stream` (hipGraph-capturable).
 *   - return value: 0 (RN_OK) or a negative rn_status; rn_last_error() has the text.
 *   - "segments": several independent problems that share one weight / one parameter set are
 *     passed as an array and run in ONE launch (the shared heads over P3..P7,
 *     reference retinanet.py:283-291).
 */
#ifndef RN_HIP_H
#define RN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rn_stream_t;

enum rn_status {
  RN_OK = 0,
  RN_EINVAL = -1,       /* bad argument */
  RN_EUNSUPPORTED = -2, /* shape outside what the kernels cover */
  RN_EHIP = -3,         /* HIP runtime error at launch */
  RN_EWORKSPACE = -4    /* workspace too small */
};

enum rn_act { RN_ACT_NONE = 0, RN_ACT_RELU = 1, RN_ACT_ELU = 2, RN_ACT_RELU6 = 3, RN_ACT_SIGMOID = 4 };
enum rn_loss_mode { RN_LOSS_BCE_DICE = 0, RN_LOSS_FOCAL = 1 };
enum rn_opt { RN_OPT_MOMENTUM = 0, RN_OPT_RMSPROP = 1, RN_OPT_ADAM = 2 };

#define RN_MAX_SEG 16

/* caller-owned list that collects deferred row reductions (see "deferred gradient reductions" below) */
typedef struct rn_reduce_list rn_reduce_list;

/* Version of this header's ABI: bumped whenever an entry point's arguments or a struct layout change.  rn_version() returns
 * the value the library was built with; a caller built against another value must not call anything else. */
#define RN_API_VERSION 407
int rn_version(void);
const char* rn_last_error(void);

/* TF 'SAME' padding rule: out = ceil(n/s); before = max((out-1)s+k-n,0)/2. */
void rn_same_pad(int n, int k, int s, int* out, int* pad_before);

/* ------------------------------------------------------------------ dense convolution
 * Replaces tf.layers.Conv2D (Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter kernels):
 * retinanet.py:39-46,55-62,87-94,100-106,127-145,170-201; mobilenet_v2.py:57-59,75-77,
 * 112-114,179-181; resnet.py:32-49,67-69,147-149; densenet.py:37,61-70,137,168.
 * padding='same' always (the reference never uses 'valid'); implicit GEMM on
 * v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate.
 */
typedef struct rn_conv_seg {
  const float* x;    /* fwd,wgrad: input  [n,h,w,cin]                       */
  const float* wgt;  /* fwd,dgrad: kernel [kh,kw,cin,cout]                  */
  const float* bias; /* fwd: [cout] or NULL                                 */
  float* y;          /* fwd: output [n,oh,ow,cout]                          */
  const float* dy;   /* dgrad,wgrad: grad of output [n,oh,ow,cout]          */
  float* dx;         /* dgrad: grad of input [n,h,w,cin]                    */
  int32_t n, h, w;   /* input batch / height / width                        */
  int32_t cout;      /* may differ per segment in fwd/dgrad                 */
  int32_t x_ld;      /* 0: x (and dx) are dense [n,h,w,cin].  > 0: x / dx are the channel slice
                        [x_coff, x_coff+cin) of a buffer with x_ld channels per pixel (x, dx point at the
                        buffer's first element): lets two convs consume / fill halves of one tensor      */
  int32_t x_coff;
  int64_t wgt_bytes; /* fp16 convs (rn_conv2d_fwd_f16*): bytes of the packed-kernel buffer `wgt` points at.  The fragment-ordered
                        copy behind Wt is read only when this proves it is there (>= rn_pack_weights_f16_bytes); 0 = unknown:
                        a Wt-only buffer is assumed.  Ignored by the fp32 entry points. */
} rn_conv_seg;

typedef struct rn_conv_geom {
  int32_t kh, kw, stride, cin;
  int32_t groups; /* 0 or 1 = dense.  G > 1: grouped conv (ResNeXt cardinality, resnet.py:53-59): kernel
                     [kh,kw,cin/G,cout], output channels [g*cout/G,(g+1)*cout/G) read input channels
                     [g*cin/G,(g+1)*cin/G) */
} rn_conv_geom;

/* workspace: optional scratch for split-K (rn_conv2d_*_workspace bytes, 0 when the launch is large enough not to
 * want it): grids of a few tiles with a long reduction -- the stride-2 convs that make P6 / P7, the 4x4 and 8x8
 * pyramid maps -- are otherwise latency-bound on a handful of CUs.  NULL / too small => no split (never an error). */
size_t rn_conv2d_fwd_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g);
size_t rn_conv2d_dgrad_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g);
int rn_conv2d_fwd(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                  rn_stream_t stream);
int rn_conv2d_dgrad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                    rn_stream_t stream);
/* ------------------------------------------------------------------ GroupNorm statistics from the producer
 * normalization.py:30 takes the moments of a conv output that this library has just had in registers: the conv /
 * depthwise forward can emit partial sums as a by-product -- per block, a row of (sum, sum of squares) pairs -- and the
 * GroupNorm that follows (rn_gn_params.stat_rows) merges the rows of its sample in a fixed order (fp64) while its
 * activations are in flight, then applies: one elementwise pass instead of a reduction + exchange + apply.  No block
 * waits for another, no atomics, bitwise reproducible.
 *   rows            : [n][rows_per_sample][width] pairs of floats, written by the producer, read by the GroupNorm
 *   rows_per_sample : m-tiles of a sample (conv) / pixel chunks of a sample (depthwise)
 *   per_group       : 0: width = channels (the conv's tiles cut through groups); 1: width = groups (a depthwise block owns
 *                     all channels of its pixels and folds them into groups itself)
 * The *_stats_rows functions return the bytes of `rows` and fill the layout fields, or return 0 when the shape cannot
 * produce rows (several segments, grouped conv, bias, split-K plan, a sample's pixels not a multiple of the tile height)
 * or the GroupNorm could not merge them cheaply (rn_group_norm_rows_ok): use the stand-alone GroupNorm then. */
typedef struct rn_gn_rows {
  void* rows;
  int32_t rows_per_sample, per_group, groups;
} rn_gn_rows;
size_t rn_conv2d_stats_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, size_t workspace_bytes, int groups,
                            rn_gn_rows* layout);
/* rn_conv2d_fwd + the rows of y (one dense segment, no bias) */
int rn_conv2d_fwd_stats(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, void* workspace, size_t workspace_bytes,
                        const rn_gn_rows* rows, rn_stream_t stream);

/* Conv2D -> Dropout (-> the statistics of the GroupNorm behind it) as ONE launch: y = dropout(conv(x)), and `rows` (may be NULL) = the
 * partial sums of the DROPPED y in the layout rn_conv2d_dropout_rows reports -- DenseNet's composite function runs
 * 1x1 conv -> Dropout -> GroupNorm (densenet.py:61-67, 70-77, dropout: densenet.py:23): the conv's output is never stored un-dropped and
 * the GroupNorm does not read it twice.  The mask is rn_dropout's (keep element i of y iff hash(seed + *seed_dev, i) >= rate, scaled by
 * 1 / (1 - rate)): bit-identical to rn_conv2d_fwd followed by rn_dropout.  Dense 1x1 / stride-1 convs of one segment without bias, in
 * product mode 1 (rn_set_product_mode): rn_conv2d_dropout_rows sets *fused_ok = 0 and the call returns RN_EUNSUPPORTED elsewhere -- the
 * caller then runs rn_conv2d_fwd + rn_dropout.  rn_conv2d_dropout_rows returns the rows' bytes (0: no rows for this shape). */
size_t rn_conv2d_dropout_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int groups, rn_gn_rows* layout, int* fused_ok);
int rn_conv2d_fwd_dropout(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float rate, uint64_t seed, const uint64_t* seed_dev,
                          const rn_gn_rows* rows, rn_stream_t stream);

/* dw[kh,kw,cin,cout] = sum over all segments (they share the kernel: shared heads);
 * split-K partial slabs go to `workspace`, reduced in fixed order (bitwise reproducible).
 * If accumulate != 0 the result is added to dw instead of overwriting it. */
size_t rn_conv2d_wgrad_workspace(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g);
int rn_conv2d_wgrad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, int accumulate,
                    void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);

/* Both gradients of one convolution (segments: x, wgt, dy, dx as for the two calls above; dw overwritten).  Small
 * problems -- the backbone's 1x1 convs -- run as ONE launch whose blocks are of two kinds (data-gradient tiles and
 * weight-gradient splits); anything else falls back to rn_conv2d_dgrad (without split-K) + rn_conv2d_wgrad.
 * workspace: rn_conv2d_wgrad_workspace bytes. */
int rn_conv2d_bwd(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dw, void* workspace,
                  size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);

/* dbias[cout] = sum over all segments / pixels of dy (the out_conv biases, retinanet.py:46-53). */
size_t rn_conv2d_bias_grad_workspace(int cout);
int rn_conv2d_bias_grad(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, float* dbias, void* workspace,
                        size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);

/* ------------------------------------------------------------------ deferred gradient reductions
 * Every weight-gradient entry point (rn_conv2d_wgrad, rn_conv2d_bwd, rn_depthwise_wgrad, rn_depthwise_bwd,
 * rn_conv2d_bias_grad, the GroupNorm parameter gradients of rn_group_norm_bwd, rn_reduce_rows) ends with a
 * fixed-order row reduction of per-block partial results.  A training step has ~130 of them, each a
 * launch-latency-bound kernel.  Each of those entry points therefore takes a trailing `rn_reduce_list* defer`:
 * NULL = reduce now; otherwise the reduction is only RECORDED in the caller's list (host memory the caller owns:
 * the library keeps nothing between calls) and rn_flush_reductions(list, stream) runs everything recorded as one
 * launch (per 140) and empties the list.  While deferring, the `workspace` handed to those entry points must stay
 * untouched until the flush (give each call its own buffer), and the gradients are only valid after the flush.
 * Results are bitwise identical to immediate mode, except the GroupNorm dgamma / dbeta of maps too large for the
 * single-kernel path, whose chunk rows are then summed in fp32 (fixed order) instead of fp64.
 * A full list is an error (RN_EWORKSPACE), never a silent immediate launch. */
typedef struct rn_reduce_desc {
  const float* in;   /* [nrows][count] partial rows */
  float* out;        /* [count] */
  int32_t count;
  uint16_t nrows, accumulate;
} rn_reduce_desc;
struct rn_reduce_list {
  rn_reduce_desc* desc; /* caller-owned array of `capacity` entries */
  int32_t capacity;
  int32_t count;        /* entries recorded so far */
};
int rn_flush_reductions(rn_reduce_list* list, rn_stream_t stream);

/* ------------------------------------------------------------------ Winograd F(m x m, 3x3) convolution
 * The 3x3 / stride-1 / SAME dense convolutions of the head towers and FPN merges (retinanet.py:39-46,87-94,
 * 138-145) with 2.25x (tile = 2) or 4x (tile = 4) fewer multiply-adds: input transform -> (tile+2)^2 batched
 * GEMMs on the fp32 matrix cores -> output transform (+ bias).  Segment fields as rn_conv2d_fwd (x, y, n, h, w;
 * dense NHWC).  w is always the forward kernel [3,3,cin,cout]; with dgrad != 0 the call computes
 * dx = conv(dy, rot180(w)^T) from the segments' dy / dx instead (the weight gradient keeps the direct
 * rn_conv2d_wgrad).  cin and cout multiples of 4.  Results differ from the direct kernels only by the fp32
 * rounding of the transforms (about 1e-6 of the output range for tile 2, 1e-5 for tile 4; tests: <= 1e-4). */
size_t rn_conv3x3_winograd_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile);
/* v_buf / urot_buf (optional, caller-owned, sizes from rn_conv3x3_winograd_keep_bytes) let a training step reuse two
 * intermediates instead of recomputing them in the backward pass: a forward call (dgrad == 0) writes the transformed
 * input into v_buf (rn_conv3x3_winograd_wgrad then skips its input transform) and the transformed ROTATED kernel into
 * urot_buf (same launch as the forward kernel transform); a data-gradient call (dgrad != 0) that is handed urot_buf
 * skips its weight transform.  NULL = compute everything in the workspace. */
int rn_conv3x3_winograd_keep_bytes(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile, size_t* v_bytes,
                                   size_t* urot_bytes);
int rn_conv3x3_winograd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, const float* bias,
                        int dgrad, int tile, void* workspace, size_t workspace_bytes, float* v_buf, float* urot_buf,
                        rn_stream_t stream);
/* The batched product the Winograd stages run on the fp32 matrix cores, exposed for measurement (bench.py times
 * the head-tower layer's instance: 36 x [682 x 256] x [256 x 256]) and reuse:
 *   C_b [M x N] = A_b [M x K] * B_b,  b = 0..nbatch-1, matrices of a batch stored back to back;
 *   b_nk == 0: B_b is [K x N];  b_nk != 0: B_b is [N x K] (the data-gradient layout).  K and N multiples of 4. */
int rn_gemm_batched(const float* A, const float* B, float* C, int M, int K, int N, int nbatch, int b_nk, rn_stream_t stream);
/* The merged backward products of such a layer (what rn_conv3x3_winograd_bwd launches between its transforms), exposed
 * for measurement (bench.py times the head-tower instance: the largest kernel of the training step): ONE launch with
 * blocks of two kinds -- the data-gradient products Cd_b [M x Nd] = Ad_b [M x Kd] * Bd_b^T (Bd_b stored [Nd x Kd]) and
 * the split partial products of the weight gradient Aw_b^T [Kw x M] * Bw_b [M x Nw], left in `workspace` as *nsplit
 * slabs (rn_winograd_bwd_products_workspace bytes). */
size_t rn_winograd_bwd_products_workspace(int M, int Kw, int Nw, int nbatch);
int rn_winograd_bwd_products(const float* Ad, const float* Bd, float* Cd, int M, int Kd, int Nd, const float* Aw,
                             const float* Bw, int Kw, int Nw, int nbatch, void* workspace, size_t workspace_bytes,
                             int* nsplit, rn_stream_t stream);
/* Weight gradient of the same convolution in the Winograd domain: dU_xi = sum over tiles of
 * (B^T d B)_xi^T (A dY A^T)_xi as (tile+2)^2 batched GEMMs with the reduction split across blocks (fixed-order
 * sum), then dw[3,3,cin,cout] (+)= G^T dU G.  Segments: x, dy, n, h, w. */
size_t rn_conv3x3_winograd_wgrad_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile);
int rn_conv3x3_winograd_wgrad(const rn_conv_seg* segs, int nseg, int cin, int cout, float* dw, int accumulate, int tile,
                              void* workspace, size_t workspace_bytes, const float* v_buf, rn_stream_t stream);

/* The whole backward pass of such a layer -- dx for every segment and dw -- in three launches whose blocks are of two
 * kinds each: [B^T dy B | A dy A^T], [data-gradient products | weight-gradient partial products],
 * [output transform -> dx | G^T dU G -> dw].  Same results as rn_conv3x3_winograd(dgrad = 1) followed by
 * rn_conv3x3_winograd_wgrad.  Segments: x, dy, dx.  v_buf / urot_buf: the buffers a forward call kept (either may be
 * NULL: then it is rebuilt here; say so in the workspace query). */
size_t rn_conv3x3_winograd_bwd_workspace(const rn_conv_seg* segs, int nseg, int cin, int cout, int tile, int have_v, int have_urot);
int rn_conv3x3_winograd_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate, int tile,
                            void* workspace, size_t workspace_bytes, const float* v_buf, const float* urot_buf, rn_stream_t stream);

/* How the batched fp32 products of the Winograd convolutions are evaluated (process-wide; default 1, or the environment's
 * RN_PROD_X3 at first use):
 *   0  the exact fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32): a k-ordered fmaf chain
 *   1  every fp32 operand split exactly into three bf16 values, six bf16 matrix-core products with fp32 accumulation
 *      (csrc/gemm_x3.hip): relative error <= ~2^-23 per elementary product, 2.7 x less matrix-core time.  Storage stays fp32.
 * Shapes the split kernels do not take (K or N not a multiple of 4, matrices >= 2 GiB) use mode 0 regardless. */
int rn_set_product_mode(int mode);
int rn_get_product_mode(void);
/* Product mode 1, the kernel operand of the head towers' FORWARD products (the reference's conv kernels of retinanet.py:37-62,85-106)
 * pre-split by its PRODUCER: the Winograd kernel transform of rn_conv3x3_winograd_gn writes U as the three bf16 planes the product
 * kernel needs, in the order its matrix-core instruction reads them -- [32-column block][16-k step][plane][lane][8 bf16], 6 bytes
 * per element -- and the product kernel loads its B fragments straight from global memory (no split, no LDS for that operand).
 * Same values, same order of operations: results are bit-identical to the fp32-operand kernel.  Process-wide switch (default on,
 * RN_X3_BFRAG=0 at first use: off).
 *   rn_x3_bfrag_ok     1 when [M x K] x [K x N] can take such an operand now (mode 1, switch on, K % 16 == 0, N % 4 == 0)
 *   rn_x3_bfrag_bytes  bytes of the images of `nbatch` [K x N] operands (the columns padded to 32)
 *   rn_x3_pack_bfrag   fp32 B_b ([K x N], or [N x K] when b_nk) -> images: for stand-alone products (tests, bench.py)
 *   rn_gemm_batched_bfrag   rn_gemm_batched with B given as images (fwd_name: which of two identical kernel instantiations runs) */
int rn_set_x3_bfrag(int on);
int rn_get_x3_bfrag(void);
int rn_x3_bfrag_ok(int M, int K, int N);
size_t rn_x3_bfrag_bytes(int K, int N, int nbatch);
int rn_x3_pack_bfrag(const float* B, void* out, int K, int N, int nbatch, int b_nk, rn_stream_t stream);
int rn_gemm_batched_bfrag(const float* A, const void* Bfrag, float* C, int M, int K, int N, int nbatch, int fwd_name, rn_stream_t stream);

/* ------------------------------------------------------------------ fp16 inference convolution
 * BASELINE configs[4] ("Inference-only ResNeXt-50-FPN 1024x1024 bs=16, fp16"): forward conv on the f16
 * matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate).  Segment fields as rn_conv2d_fwd but x is
 * fp16 NHWC, wgt is the PACKED kernel (fp16, written by rn_pack_weights_f16 into a buffer of rn_pack_weights_f16_bytes
 * bytes), bias fp32, y fp16 (out_f32 = 0) or fp32 (out_f32 = 1).  cin/G must be a multiple of 4.
 * The packed kernel is Wt[cout][kh*kw*cin/G] and, when kh*kw*cin/G is a multiple of 16, 256-byte aligned behind it, the
 * same values once more in matrix-core fragment order Wf[ceil(cout/32)][K/16][64 lanes][8] (channels padded with zeros):
 * the large-tile kernel reads its weight operand from that copy straight into registers.
 * A 3 x 3 kernel with 4 / 8 / 16 / 32 input channels per group and cout % 32 == 0 carries a THIRD copy, 256-byte aligned behind
 * the others: Ws[cout/32][9 taps][2][64 lanes][8], the block-diagonal 32 x 32 kernel of every super-group of 32 consecutive
 * channels in fragment order (lane l: output channel l & 31, input channels 16 step + 8 (l >> 5) .. + 7 of the super-group; zero
 * where the two belong to different groups of cin_g channels).  Grouped 3 x 3 convs with as many input as output channels per
 * group on maps of whole 16 x 16 (stride 1) / 32 x 32 (stride 2) pixel tiles -- ResNeXt's conv 2, resnet.py:36-49 -- run from it:
 * a block keeps the input patch of a 16 x 16 (8 x 16) output tile of one super-group in LDS and the kernel in registers; with
 * rn_conv2d_fwd_f16_fold a GroupNorm + activation in front is applied once per patch element on its way into LDS. */
size_t rn_pack_weights_f16_bytes(int kh, int kw, int cin_g, int cout);
int rn_pack_weights_f16(const float* w, void* wt, int kh, int kw, int cin_g, int cout, rn_stream_t stream);
int rn_cast_f32_to_f16(const float* x, void* y, int64_t count, rn_stream_t stream);
/* image [pixels,3] fp32 -> [pixels,4] fp16 with a zero 4th channel (the stem then gathers 8 bytes per tap) */
int rn_pad_cast_rgb_f16(const float* x, void* y, int64_t pixels, rn_stream_t stream);
int rn_conv2d_fwd_f16(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int out_f32, rn_stream_t stream);
/* The same convolution with the GroupNorms around it folded in (Sequential([Conv2D, Normalization, activation]) of the
 * backbones, e.g. resnet.py:53-64,88-101, normalization.py:20-35), fp16 output, one segment:
 *   input side (in_mean != NULL): x is the RAW output of the previous conv; the kernel loads act(GN(x)) -- scale / shift per
 *     (sample, channel) from in_mean / in_rstd [n][in_groups] and in_gamma / in_beta [cin] -- and pads with zeros AFTER
 *     the activation, exactly what a stand-alone GroupNorm + activation pass would have stored;
 *   output side (partial != NULL): per (m-tile, channel) sums of y and y^2, of the values as stored in fp16, in the
 *     layout [2][n * rows][cout] with rows = rn_conv2d_f16_fold_rows(...) m-tiles per sample; rn_group_norm_finalize turns
 *     them into mean / rstd [n][groups]; the conv must have no bias.  (The input side alone is not built: a conv whose
 *     input is a pending GroupNorm is followed by one itself in every network here.)
 * rn_conv2d_f16_fold_rows returns 0 when the shape cannot fold (several segments, fp32 output, oh*ow not a whole number of
 * m-tiles, cout/groups not a multiple of 8): the caller then uses rn_conv2d_fwd_f16 + rn_group_norm_fwd. */
typedef struct rn_f16_fold {
  const float* in_mean; const float* in_rstd; const float* in_gamma; const float* in_beta;
  int in_groups; int in_act;
  float* partial;
  /* several segments in one launch (the head towers' pyramid levels, retinanet.py:37-62,85-106; output side only): `partial` is ONE
   * array [2][total_chunks][cout]; segment s writes its rows (n_s x tiles per sample, rn_conv2d_f16_stats_tiles) from row
   * seg_chunk_start[s] on; a segment with 0 tiles per sample (its m-tiles straddle samples) writes none.  NULL: one segment, as above. */
  const int32_t* seg_chunk_start;
  int32_t total_chunks;
} rn_f16_fold;
int rn_conv2d_f16_fold_rows(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g);
/* m-tiles per sample each segment of such a launch would write statistics rows for (0: none) */
int rn_conv2d_f16_stats_tiles(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, int32_t* tiles_per_sample);
int rn_conv2d_fwd_f16_fold(const rn_conv_seg* segs, int nseg, const rn_conv_geom* g, const rn_f16_fold* fold, rn_stream_t stream);
/* mean / rstd [n][groups] from the partial sums above (fp64, fixed order); hw = oh * ow */
int rn_group_norm_finalize(const float* partial, int n, int rows_per_sample, int hw, int c, int groups, float eps, float* mean,
                           float* rstd, rn_stream_t stream);
/* y = act(GN(x)) + residual, or act(GN(x) + residual) (act_after_residual), fp16 x / residual / y, from given statistics */
int rn_group_norm_apply_f16(const void* x, const void* residual, void* y, int n, int hw, int c, int groups, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int act, int act_after_residual,
                            rn_stream_t stream);
/* The same with a residual that is itself the RAW fp16 output of a conv whose activation-free GroupNorm has not been applied yet
 * (ResNeXt's projection branch, resnet.py:62-71: identity = normalization(conv(input))): residual' = GN_r(residual) is formed
 * in fp32 inside this pass and never written -- y = act(GN(x) + GN_r(residual)) (act_after_residual) or act(GN(x)) + GN_r(..). */
typedef struct rn_gn_residual_norm {
  const float* mean; const float* rstd;       /* [n][groups] */
  const float* gamma; const float* beta;      /* [c] */
  int groups;
} rn_gn_residual_norm;
int rn_group_norm_apply_res_f16(const void* x, const void* residual, const rn_gn_residual_norm* residual_norm, void* y, int n, int hw, int c,
                                int groups, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                int act_after_residual, rn_stream_t stream);

/* ------------------------------------------------------------------ depthwise 3x3
 * Replaces tf.nn.depthwise_conv2d (DepthwiseConv2dNative + its two backprops),
 * mobilenet_v2.py:35-36.  Kernel [kh,kw,C] (channel multiplier 1), padding SAME.
 */
int rn_depthwise_fwd(const float* x, const float* wgt, float* y, int n, int h, int w, int c, int k, int stride,
                     rn_stream_t stream);
/* the same with the GroupNorm partial-sum rows of y (rn_gn_rows above); k == 3, c % 4 == 0, c <= 1024 */
size_t rn_depthwise_stats_rows(int n, int h, int w, int c, int k, int stride, int groups, rn_gn_rows* layout);
int rn_depthwise_fwd_stats(const float* x, const float* wgt, float* y, int n, int h, int w, int c, int k, int stride,
                           const rn_gn_rows* rows, rn_stream_t stream);
int rn_depthwise_dgrad(const float* dy, const float* wgt, float* dx, int n, int h, int w, int c, int k,
                       int stride, rn_stream_t stream);
size_t rn_depthwise_wgrad_workspace(int n, int h, int w, int c, int k, int stride);
/* both gradients of a 3x3 depthwise conv in one launch (workspace: rn_depthwise_wgrad_workspace bytes) */
int rn_depthwise_bwd(const float* x, const float* dy, const float* wgt, float* dx, float* dw, int n, int h, int w, int c, int k,
                     int stride, void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);
int rn_depthwise_wgrad(const float* x, const float* dy, float* dw, int n, int h, int w, int c, int k, int stride,
                       void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);

/* ------------------------------------------------------------------ fused GroupNorm -> depthwise 3x3 -> GroupNorm
 * The middle of a MobileNetV2 bottleneck, mobilenet_v2.py:56-80 -- Sequential([.., Normalization(), activation, Dropout]),
 * DepthwiseConv2D(3, strides), Normalization(), activation, Dropout --
 *     y = drop2(act(GN2( depthwise3x3_same( drop1(act(GN1(x))) ) )))
 * as ONE kernel forward and ONE backward instead of three each.  A depthwise conv keeps channels apart and both
 * GroupNorms group the same channels, so one block owns a (sample, group) slice end to end, in LDS: exact two-pass
 * statistics, no exchange between blocks, the two intermediate tensors are never written (backward recomputes them
 * from x).  Applies when a slice fits a CU's LDS (rn_dwgn_supported: forward needs h*w*(c/groups)*4 bytes, backward the
 * input and output slices together, <= 156 KB) -- MobileNetV2 at 1/8 resolution and below for a 512^2 image; otherwise
 * the caller runs the three stand-alone kernels.  The dropouts hash (drop_seedK + *drop_seed_dev, element index) exactly
 * like rn_group_norm_fwd, so fused and unfused results agree to rounding.
 */
typedef struct rn_dwgn_params {
  int32_t n, h, w, c;      /* input [n,h,w,c]; output [n,ceil(h/stride),ceil(w/stride),c] */
  int32_t stride;          /* 1 or 2 */
  int32_t groups, act;     /* of both GroupNorms */
  float eps, drop_rate;
  uint64_t drop_seed1, drop_seed2;
  const uint64_t* drop_seed_dev;
} rn_dwgn_params;
int rn_dwgn_supported(const rn_dwgn_params* p, int backward);
/* stats [4][n][groups]: mean1, rstd1, mean2, rstd2 (kept for the backward pass) */
int rn_dwgn_fwd(const float* x, const float* gamma1, const float* beta1, const float* wgt, const float* gamma2,
                const float* beta2, float* y, float* stats, const rn_dwgn_params* p, rn_stream_t stream);
/* dx [n,h,w,c]; rows: per-sample partial parameter gradients, to be summed over the n rows of each section with
 * rn_reduce_rows: [dgamma1: n x c][dbeta1: n x c][dgamma2: n x c][dbeta2: n x c][dw: n x 9c] */
int rn_dwgn_bwd(const float* x, const float* dy, const float* gamma1, const float* beta1, const float* wgt,
                const float* gamma2, const float* beta2, const float* stats, float* dx, float* rows,
                const rn_dwgn_params* p, rn_stream_t stream);

/* ------------------------------------------------------------------ GroupNorm (+act, +dropout, +residual)
 * Replaces normalization.py:20-35 (reshape + tf.nn.moments + affine), the activation that
 * follows it in every reference Sequential (tf.nn.elu train.py:214 / tf.nn.relu resnet.py:85),
 * tf.layers.Dropout (mobilenet_v2.py:62,71,79) and the residual add (mobilenet_v2.py:91-92).
 *   y = dropout(act((x-mean_g)*rstd_g*gamma_c+beta_c)) + residual
 * Segments share gamma/beta (heads: one layer applied to P3..P7).
 * mean / rstd ([n, groups] per segment) are outputs of fwd and inputs of bwd.
 */
typedef struct rn_gn_seg {
  const float* x;        /* [n, hw, c] conv output                         */
  float* y;              /* fwd out                                         */
  const float* residual; /* fwd: optional, same shape as y, or NULL         */
  const float* dy;       /* bwd in                                          */
  float* dx;             /* bwd out                                         */
  float* dresidual;      /* bwd out, only with act_after_residual: grad w.r.t. residual */
  float* mean;           /* [n, groups]                                     */
  float* rstd;           /* [n, groups]                                     */
  int32_t n, hw;
  /* Channel-prefix views (concat-free DenseNet blocks, densenet.py:117-121: layer i normalises the first c channels of
   * the block's one [n, hw, c_total] buffer instead of a concatenated copy).  0 = dense. */
  int32_t x_ld;          /* floats between consecutive pixels of x (fwd and bwd reads)                     */
  int32_t dx_ld;         /* the same for dx                                                                 */
  int32_t dx_accumulate; /* != 0: dx += (every later layer of the block adds to the same gradient buffer)   */
} rn_gn_seg;

typedef struct rn_gn_params {
  int32_t c, groups, act;
  int32_t act_after_residual; /* 0: y = drop(act(GN(x))) + residual (MobileNetV2, mobilenet_v2.py:91-92);
                                 1: y = drop(act(GN(x) + residual))  (ResNeXt, resnet.py:99-101)          */
  int32_t in_f16, out_f16;    /* forward only (inference): x is fp16 / y and residual are fp16           */
  float eps;
  float drop_rate;    /* 0 => no dropout                                    */
  uint64_t drop_seed; /* counter-based mask: keep iff hash(seed, elem) >= rate            */
  const uint64_t* drop_seed_dev; /* optional DEVICE counter added to drop_seed (so a replayed
                                    hipGraph draws a fresh mask every step); may be NULL    */
  void* sync; /* optional: rn_group_norm_sync_bytes() zero-initialised DEVICE bytes owned by the caller, private to the
                 stream, written by these kernels only; enables the single-kernel path for mid-sized maps (<= 128 blocks
                 exchange tagged per-group sums through it, bounded polling).  Word 0 is left at zero, word 1 counts the
                 calls, word 2 becomes 1 if a wait ever timed out.  NULL = not used. */
  const rn_gn_rows* stat_rows; /* forward, one dense fp32 segment: the partial-sum rows its producer wrote for x
                                  (rn_conv2d_fwd_stats / rn_depthwise_fwd_stats): merged here instead of reading x twice.
                                  Ignored (statistics computed from x) where rn_group_norm_rows_ok says no.  NULL = none. */
} rn_gn_params;

size_t rn_group_norm_sync_bytes(void);
/* can rn_group_norm_fwd merge rows of this layout for c channels in `groups` groups? (channel slabs of whole groups must
 * tile c, and a block's share of the rows must stay small) */
int rn_group_norm_rows_ok(int c, int groups, int rows_per_sample, int per_group);
size_t rn_group_norm_workspace(const rn_gn_seg* segs, int nseg, const rn_gn_params* p);
int rn_group_norm_fwd(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma,
                      const float* beta, void* workspace, size_t workspace_bytes, rn_stream_t stream);
/* fp16 inference, several segments (the head towers' pyramid levels: the GroupNorm + activation behind the tower convs of
 * retinanet.py:37-62,85-106): y = act(GN(x)) with the statistics taken from the conv's epilogue rows (rn_conv2d_fwd_f16_fold with
 * seg_chunk_start; `partial` and `tiles_per_sample` as there / as rn_conv2d_f16_stats_tiles returned them) instead of a pass over x;
 * a segment with 0 tiles per sample (<= 4096 pixels per sample) is summed from x by the finalise blocks.  fp16 x and y, dense,
 * c % 64 == 0, no dropout, no residual.  Two launches (finalise, apply). */
int rn_group_norm_fwd_f16_tiles(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma, const float* beta,
                                const float* partial, const int32_t* tiles_per_sample, rn_stream_t stream);
/* dgamma/dbeta [c] are OVERWRITTEN with the sum over all segments. */
int rn_group_norm_bwd(const rn_gn_seg* segs, int nseg, const rn_gn_params* p, const float* gamma,
                      const float* beta, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                      rn_stream_t stream, rn_reduce_list* defer);

/* ------------------------------------------------------------------ small elementwise ops
 * activation alone (retinanet.py:180-181 `activation` before the P7 conv) */
int rn_act_fwd(const float* x, float* y, int64_t count, int act, rn_stream_t stream);
int rn_act_bwd(const float* x, const float* dy, float* dx, int64_t count, int act, rn_stream_t stream);
/* y = lateral + nearest_resize(top -> lateral size, align_corners=True); retinanet.py:153-157
 * (ResizeNearestNeighbor) and its gradient w.r.t. top (sum over the children). */
int rn_upsample_add_fwd(const float* lateral, const float* top, float* y, int n, int h, int w, int th, int tw, int c,
                        rn_stream_t stream);
int rn_upsample_add_bwd_top(const float* dy, float* dtop, int n, int h, int w, int th, int tw, int c,
                            rn_stream_t stream);

/* stand-alone inverted dropout (DenseNet puts tf.layers.Dropout after a conv: densenet.py:44,67,77,143);
 * the same counter-based mask in forward and backward: dx = rn_dropout(dy) with the same seed.
 *
 * THE MASK FUNCTION (every dropout of this library: rn_dropout*, rn_gn_params, rn_dwgn_params, rn_mb_norm).  Element e (flat
 * index into the dense NHWC tensor the dropout acts on) with s = seed + (seed_dev ? *seed_dev : 0), all arithmetic mod 2^32:
 *     h  = lo32(e) * 0x9E3779B1 + lo32(s);   h ^= hi32(e) * 0x85EBCA77 + hi32(s);
 *     h ^= h >> 16;  h *= 0x85EBCA6B;  h ^= h >> 13;  h *= 0xC2B2AE35;  h ^= h >> 16;
 *     u  = (float)(h >> 8) * 2^-24;          y = u >= rate ? x * (1.f / (1.f - rate)) : 0
 * Nothing else enters it (not the launch shape, not the kernel that applies it), so a checker can be handed the very masks
 * a step draws: oracle/dropout_ref.py restates it, tests/test_gpu_dropout.py compares the two bit for bit. */
int rn_dropout(const float* x, float* y, int64_t count, float rate, uint64_t seed, const uint64_t* seed_dev,
               rn_stream_t stream);
/* The same between channel slices: y[p, y_coff + ch] = dropout(x[p, x_coff + ch]), ch < c, rows x_ld / y_ld floats apart; the
 * mask is that of the dense [pixels, c] tensor (element p * c + ch), so slice and dense forms agree.  rate 0 = a copy.
 * Concat-free DenseNet blocks (densenet.py:117-121): a growth layer's output goes straight into its channel slice of the
 * block's buffer; in the backward pass its gradient slice is read out of the block's gradient buffer. */
int rn_dropout_strided(const float* x, float* y, int64_t pixels, int c, int x_ld, int x_coff, int y_ld, int y_coff, float rate,
                       uint64_t seed, const uint64_t* seed_dev, rn_stream_t stream);
/* tf.layers.MaxPooling2D(k, stride, 'same') (resnet.py:200, densenet.py:180): padded cells never win; the
 * gradient goes to the first maximum of each window.  tf.layers.AveragePooling2D(k, stride, 'same')
 * (densenet.py:144): divides by the number of valid cells. */
/* argmax (optional, uint8 [n,oh,ow,c]): tap index kh*k+kw of each window's first maximum, for rn_maxpool_bwd_arg */
int rn_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int n, int h, int w, int c, int k, int stride,
                   rn_stream_t stream);
/* fp16-storage forward variants for the inference path (same semantics) */
int rn_act_fwd_f16(const void* x, void* y, int64_t count, int act, rn_stream_t stream);
int rn_maxpool_fwd_f16(const void* x, void* y, int n, int h, int w, int c, int k, int stride, rn_stream_t stream);
/* fp16 inference: y = maxpool_kxk/stride(act(GroupNorm(x))) from the RAW conv output x [n,h,w,c] (fp16) and the GroupNorm's
 * per-(sample, group) mean / rstd (rn_group_norm_finalize): ResNeXt's stem (resnet.py:192-201: conv, normalization, relu,
 * max_pooling2d) without writing the normalised tensor.  Bit-equal to rn_group_norm_apply_f16 followed by rn_maxpool_fwd_f16. */
int rn_maxpool_gn_fwd_f16(const void* x, void* y, int n, int h, int w, int c, int k, int stride, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int groups, int act, rn_stream_t stream);
int rn_upsample_add_fwd_f16(const void* lateral, const void* top, void* y, int n, int h, int w, int th, int tw, int c,
                            rn_stream_t stream);
/* two forms of the same gradient: from x (re-scans every window) or from the forward pass's argmax bytes (5 bytes
 * read per window; what the autograd carrier uses) */
int rn_maxpool_bwd(const float* x, const float* dy, float* dx, int n, int h, int w, int c, int k, int stride,
                   rn_stream_t stream);
int rn_maxpool_bwd_arg(const uint8_t* argmax, const float* dy, float* dx, int n, int h, int w, int c, int k, int stride,
                       rn_stream_t stream);
int rn_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int k, int stride, rn_stream_t stream);
int rn_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int k, int stride, rn_stream_t stream);

/* augmentation.flip (augmentation.py:5-22): y[o, W-1-j, i] = x[o, j, i] for an [outer, W, inner] tensor of
 * 4-byte (fp32) or 1-byte (mask) elements; neg_mod > 0 also negates element neg_idx of every group of
 * neg_mod values along `inner` (the x shift of the [.., A, 4] regression maps: neg_mod 4, neg_idx 1). */
int rn_flip_width(const void* x, void* y, int64_t outer, int w, int64_t inner, int elem_bytes, int neg_mod, int neg_idx,
                  rn_stream_t stream);

/* Input pipeline: tf.image.convert_image_dtype (in_u8: uint8 * 1/255) -> bilinear resize with
 * align_corners=True (dataset.py:145-151 rescale_image; the TF ResizeBilinear arithmetic, operation by operation)
 * -> optional (v - mean[c]) / std[c] (train.py:48-49 preprocess_image; mean / std are HOST arrays of c floats, both
 * NULL = no normalisation).  x [n,h,w,c] uint8 or fp32, y [n,oh,ow,c] fp32, c <= 8. */
int rn_resize_bilinear_normalize(const void* x, int in_u8, float* y, int n, int h, int w, int c, int oh, int ow,
                                 const float* mean, const float* stdv, rn_stream_t stream);

/* ------------------------------------------------------------------ loss
 * Replaces utils.process_labels_and_logits/postprocess_and_mask (utils.py:240-284; the
 * boolean_mask compaction becomes a 0/1 row weight, result-identical) and losses.loss
 * (losses.py:155-175): class loss = BCE+dice (live, :124-139) or focal (:6-15,:119-122),
 * regression loss = Huber(delta 1) over fg rows / (4*#fg) (:144-152).
 * Segment = one pyramid level, rows = n*h*w*anchors.
 */
typedef struct rn_loss_seg {
  const float* cls_logit; /* [rows, C]                                      */
  const float* cls_label; /* [rows, C] one-hot / zeros                      */
  const float* reg_pred;  /* [rows, 4]                                      */
  const float* reg_label; /* [rows, 4]                                      */
  const uint8_t* trainable; /* [rows] 0/1                                   */
  float* d_cls_logit;     /* bwd out [rows, C]                              */
  float* d_reg_pred;      /* bwd out [rows, 4]                              */
  int64_t rows;
} rn_loss_seg;

/* stats layout (float32, device): [0]=class_loss [1]=regr_loss [2]=M (trainable rows)
 * [3]=#fg [4]=sum bce [5]=sum focal [6]=sum huber [7]=reserved, then per class c:
 * [8+3c]=I_c=sum l*sigmoid, [9+3c]=L_c=sum l, [10+3c]=P_c=sum sigmoid (the dice term's sums: RN_LOSS_BCE_DICE; in
 * RN_LOSS_FOCAL mode, which does not use them, they may be zero). */
#define RN_LOSS_STATS_HEADER 8
size_t rn_loss_workspace(const rn_loss_seg* segs, int nseg, int num_classes);
/* class_loss_out / regr_loss_out (optional, one float each): the two losses also as stand-alone device scalars */
int rn_loss_fwd(const rn_loss_seg* segs, int nseg, int num_classes, int mode, float* stats, float* class_loss_out,
                float* regr_loss_out, void* workspace, size_t workspace_bytes, rn_stream_t stream);
/* d(g_cls*class_loss + g_reg*regr_loss)/d(logits); g_* are device scalars (upstream grads). */
int rn_loss_bwd(const rn_loss_seg* segs, int nseg, int num_classes, int mode, const float* stats,
                const float* g_cls, const float* g_reg, rn_stream_t stream);

/* ------------------------------------------------------------------ GroupNorm folded into Winograd layers
 * A chain [conv3x3, GroupNorm, activation] x k -> conv3x3 (the class / box subnets, retinanet.py:37-71,85-115; the
 * GroupNorm is normalization.py:20-35) without GroupNorm kernels and without ever writing the normalised tensor:
 * the output transform of a layer emits, per chunk of 16 Winograd tiles of one sample, the per-group statistics
 * (count, mean, M2 = sum (y - mean)^2, unused) of the RAW conv output ("stat rows", rn_wino_gn_rows() rows of
 * [groups][4] floats); the input transform of the next layer merges its sample's rows (Chan et al.'s pairwise
 * update, fp64, chunk order: no E[y^2] - E[y]^2 cancellation) into mean / rstd and loads act(GN(y)) on the fly.  Backward likewise: the data gradient of the next layer leaves its output transform as
 * g = dA * act'(z) together with rows of (sum g, sum g xhat) -- per channel for dgamma / dbeta (rn_reduce_rows) and
 * gamma-weighted per group -- and the dy transforms of this layer load dy = rstd (gamma g - c1 - xhat c2) from g
 * and the raw y.  No exchange between blocks, no atomics: every sum runs over rows an earlier launch wrote.
 * Needs cin, cout multiples of 64 and 64 % (channels / groups) == 0 on a folded side (RN_EUNSUPPORTED otherwise).
 */
typedef struct rn_wino_gn {
  /* input side: segs[].x is the raw output of the previous conv; NULL in_rows = x is used as it is */
  const float* in_rows;   /* [rows][in_groups][4] written by the previous layer's call */
  const float* in_gamma;  /* [cin] */
  const float* in_beta;   /* [cin] */
  int32_t in_groups, in_act;
  float in_eps;
  /* output side: NULL out_rows = no statistics */
  float* out_rows;        /* [rows][out_groups][4] */
  int32_t out_groups;
  /* the kernel transforms prepared ahead of the layer by rn_conv3x3_winograd_gn_weights (train.Trainer: all tower layers once per
   * step on a side stream beside the backbone, off the layers' critical path): u_ready = its u_out (NULL: transformed inside this
   * call, in the workspace); urot_ready != 0: urot_buf already holds its urot_out and is only read */
  const void* u_ready;
  int32_t urot_ready;
} rn_wino_gn;
size_t rn_wino_gn_rows(const rn_conv_seg* segs, int nseg, int tile);
/* The kernel transforms of such a layer alone: U in the format the layer's forward product reads under the current switches
 * (fp32 [P][cin][cout], or the fragment image of rn_x3_pack_bfrag) -> u_out (rn_conv3x3_winograd_gn_u_bytes bytes), and the rotated
 * kernel's transform (fp32, the data gradient's operand) -> urot_out (urot_bytes of rn_conv3x3_winograd_keep_bytes; NULL: skipped).
 * Same kernels, same values as inside rn_conv3x3_winograd_gn.  The reference's conv kernels: retinanet.py:37-62,85-106. */
size_t rn_conv3x3_winograd_gn_u_bytes(int cin, int cout, int tile);
int rn_conv3x3_winograd_gn_weights(const float* w, int cin, int cout, int tile, void* u_out, size_t u_out_bytes, float* urot_out,
                                   rn_stream_t stream);
/* y = conv3x3_same(act(GN(x)), w) (+ bias).  Workspace / v_buf / urot_buf as rn_conv3x3_winograd. */
int rn_conv3x3_winograd_gn(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, const float* bias, int tile,
                           const rn_wino_gn* gn, void* workspace, size_t workspace_bytes, float* v_buf, float* urot_buf,
                           rn_stream_t stream);
typedef struct rn_wino_gn_bwd {
  /* input side folded (in_rows != NULL): segs[].x = raw input tensor, segs[].dx receives g of the input's GroupNorm */
  const float* in_rows;
  const float* in_gamma;
  const float* in_beta;
  int32_t in_groups, in_act;
  float in_eps;
  float* in_g_rows_group; /* out: [rows][in_groups][2] (sum gamma g, sum gamma g xhat) */
  float* in_g_rows_chan;  /* out: [2][rows][cin]: plane 0 sum g (-> dbeta), plane 1 sum g xhat (-> dgamma) */
  /* output side folded (out_rows != NULL): segs[].y = raw conv output, segs[].dy = g of the GroupNorm after it */
  const float* out_rows;
  const float* out_g_rows_group;
  const float* out_gamma;
  int32_t out_groups;
  float out_eps;
  int32_t defer_wgrad;    /* != 0: this call leaves dw alone -- its weight-gradient half (the product V^T dM over the tiles and the
                             G^T dU G back-transform, nothing of which anybody needs before the optimizer step) is run later by
                             rn_conv3x3_winograd_gn_bwd_wgrad FROM THE SAME WORKSPACE, which the caller keeps untouched until then:
                             e.g. on a side stream beside the backbone's latency-bound backward pass */
} rn_wino_gn_bwd;
/* dx (or g, see above) for every segment and dw (+)=; workspace rn_conv3x3_winograd_bwd_workspace bytes. */
int rn_conv3x3_winograd_gn_bwd(const rn_conv_seg* segs, int nseg, int cin, int cout, const float* w, float* dw, int accumulate,
                               int tile, const rn_wino_gn_bwd* gn, void* workspace, size_t workspace_bytes, const float* v_buf,
                               const float* urot_buf, rn_stream_t stream);
/* The deferred half of a rn_conv3x3_winograd_gn_bwd call made with gn->defer_wgrad: same segments, cin, cout, tile, workspace
 * (untouched since), v_buf and the same urot_buf-was-given flag; dw (+)= as `accumulate` says. */
int rn_conv3x3_winograd_gn_bwd_wgrad(const rn_conv_seg* segs, int nseg, int cin, int cout, float* dw, int accumulate, int tile,
                                     void* workspace, size_t workspace_bytes, const float* v_buf, int urot_was_given, rn_stream_t stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_r in[r * count + i], r < nrows, fixed order (bit-reproducible); joins the
 * caller's deferred batch when `defer` is given.  Finishes dgamma / dbeta from in_g_rows_chan. */
int rn_reduce_rows(const float* in, float* out, int64_t count, int nrows, int accumulate, rn_stream_t stream, rn_reduce_list* defer);

/* ------------------------------------------------------------------ MobileNetV2 bottleneck chain
 * Replaces the chain  [conv1x1, Normalization, act, Dropout] -> [DepthwiseConv2D 3x3, Normalization, act, Dropout] ->
 * [conv1x1, Normalization, Dropout] (+ input)  of mobilenet_v2.py:41-94 (GroupNorm variant: normalization.py:20-35),
 * forward and backward, with every GroupNorm applied by its CONSUMER while it loads:
 *   - a conv / depthwise kernel writes its RAW output y and, as a by-product, rows of per-group (sum, sum of squares);
 *   - the next kernel merges the rows of its sample (fixed order, fp64) and computes dropout(act(GN(y))) [+ residual] on the
 *     fly in its operand load -- the normalised tensor is never written (a pointwise conv can also write it out once,
 *     `materialise`: the bottleneck's output is needed again as the next bottleneck's residual / as a pyramid tap);
 *   - backward: a data-gradient kernel turns its result into g = d * act'(z) * dropout mask of the GroupNorm block it enters
 *     from behind, stores it, and emits rows of (sum gamma g, sum gamma g xhat) + per-channel planes (sum g, sum g xhat);
 *     the kernel that needs dy = rstd (gamma g - c1 - xhat c2) of that GroupNorm computes it while loading g and y.
 * One bottleneck = 3 launches forward, 3 backward (13 through the layer-by-layer path).  No kernel waits for another
 * block, no atomics; results are bitwise reproducible.  fp32, NHWC; channels % 4 == 0; a sample's pixels are a multiple
 * of the pointwise kernels' tile height (RN_EUNSUPPORTED otherwise: use the layer-by-layer entry points).
 *
 * rn_mb_rows: [n][rows_per_sample][width] pairs of floats.  Entry (group g, producer N-tile t) of a row sits at position
 * g + t: a pointwise conv's N-tiles (bn channels each) cut through groups, consecutive tiles share at most one group, so
 * the positions are unique; a depthwise block owns whole groups (bn >= channels: t = 0).  Group g's total is the sum over
 * the rows and over t = (g cpg) / bn .. ((g + 1) cpg - 1) / bn.  The *_rows functions fill the layout and return the bytes. */
typedef struct rn_mb_rows {
  float* rows;
  int32_t rows_per_sample, width, bn;
} rn_mb_rows;

/* one GroupNorm (+ activation + dropout) block, described for the kernels on either side of it */
typedef struct rn_mb_norm {
  const float* y;             /* the raw conv output it normalises, [n, hw, c] */
  rn_mb_rows stat;            /* forward: y's (sum, sum sq) rows; stat.rows == NULL: mean / rstd below are READ, not written */
  float* mean; float* rstd;   /* [n, groups]; the forward consumer writes them, every backward kernel reads them */
  const float* gamma; const float* beta;   /* [c] */
  int32_t c, groups, act;     /* act: rn_act applied after the normalisation */
  float eps, drop_rate;       /* dropout after the activation: keep iff hash(seed, element index of y) >= rate (rn_gn_params) */
  uint64_t drop_seed; const uint64_t* drop_seed_dev;
} rn_mb_norm;

/* A consumer block merges at most rn_mb_rows_max() rows of its sample.  The largest maps produce more (one row per tile):
 * rn_mb_compact_rows sums consecutive rows (fp64, fixed order) into <= 8 per sample first -- one small launch. */
int rn_mb_rows_max(void);
size_t rn_mb_compact_rows_layout(int n, const rn_mb_rows* in, rn_mb_rows* out);   /* fills `out`'s layout, returns its bytes */
int rn_mb_compact_rows(const rn_mb_rows* in, const rn_mb_rows* out, int n, rn_stream_t stream);

/* y[n,hw,cout] = A w, w [cin, cout] (a 1x1 Conv2D kernel, HWIO), A = x (plain) or dropout(act(GN(in->y))) [+ residual];
 * `materialise` (optional, with `in`): A is also written out, [n,hw,cin].  stat_out (optional): y's rows (layout from
 * rn_mb_pointwise_rows with the GroupNorm's `groups` that follows y). */
size_t rn_mb_pointwise_rows(int n, int hw, int cin, int cout, int groups, rn_mb_rows* layout);
int rn_mb_pointwise_fwd(const float* x, const rn_mb_norm* in, const float* residual, float* materialise, const float* w, float* y,
                        int n, int hw, int cin, int cout, const rn_mb_rows* stat_out, int stat_groups, rn_stream_t stream);
/* y = depthwise3x3(dropout(act(GN(in->y)))), TF SAME padding, stride 1 or 2; w [3,3,c]; stat_out: y's rows */
size_t rn_mb_depthwise_rows(int n, int h, int w, int c, int stride, int groups, rn_mb_rows* layout);
int rn_mb_depthwise_fwd(const rn_mb_norm* in, const float* w, float* y, int n, int h, int wd, int stride, const rn_mb_rows* stat_out,
                        int stat_groups, rn_stream_t stream);
/* out = dropout(act(GN(in->y))) [+ residual]: writes a normalised tensor out (the end of a chain; tests) */
int rn_mb_apply(const rn_mb_norm* in, const float* residual, float* out, int n, int hw, rn_stream_t stream);

/* ---- XCD-resident small-map section (csrc/mb_resident.hip): the same forward kernels as PHASES of one launch.
 * For the part of the chain whose per-sample working set fits one XCD's 4 MB L2 (maps of <= 32 x 32 pixels at the BASELINE sizes:
 * bottleneck_4_1's depthwise conv ... bottleneck_7_1 and the output conv of mobilenet_v2.py:120-223, ~33 dependent launches).
 * Grid = 8 x B blocks; the blocks with equal blockIdx.x % 8 sit on one XCD (checked at run time) and work on sample
 * blockIdx.x % 8 (+ 8 ...); between phases they meet at a same-XCD counter barrier (no cache write-back, no cross-XCD fence).
 * A phase = one rn_mb_pointwise_fwd (with `in`; ELU or no activation) or one rn_mb_depthwise_fwd (ELU) call, same meaning of
 * every field.  Statistic rows of resident phases have their OWN layouts (rn_mb_resident_rows): the tiling follows the cluster.
 * Caller's obligations: every buffer written by a phase is written by that phase only and read only by LATER phases of the
 * call; `sync` = rn_mb_resident_sync_bytes() zero-initialised device bytes, private to the stream, written by these launches
 * only.  Word 512 of `sync` is the error word: bit 0 = a barrier wait timed out (a block of the cluster never arrived; the
 * outputs of that call are invalid and the caller must zero `sync` before the next use), bit 1 = a cluster's blocks were seen
 * on two XCDs (placement is not as assumed: results may be stale; stop using the resident path on this device).
 * RN_EUNSUPPORTED: a phase cannot run resident (shape, activation) or the cluster does not fit an XCD -- nothing was launched,
 * use the launch-ordered entry points. */
enum { RN_MB_PHASE_POINTWISE = 0, RN_MB_PHASE_DEPTHWISE = 1 };
typedef struct rn_mb_phase {
  int32_t kind;
  const rn_mb_norm* in;        /* the GroupNorm block this phase consumes while loading (rows from the previous phase / kernel) */
  const float* residual;       /* pointwise after a linear block: + residual (or NULL) */
  float* materialise;          /* pointwise: the formed operand written out once (or NULL) */
  const float* w; float* y;    /* pointwise: w [cin, cout], y [n, h wd, cout]; depthwise: w [3, 3, c], y [n, oh, ow, c] */
  int32_t h, wd, cin, cout, stride;   /* the INPUT map; depthwise: cin == cout, stride 1 / 2; pointwise: stride ignored */
  rn_mb_rows stat_out;         /* y's rows (layout: rn_mb_resident_rows); stat_out.rows == NULL: none */
  int32_t stat_groups;
} rn_mb_phase;
size_t rn_mb_resident_sync_bytes(void);
/* layout + bytes of the rows a resident phase of this shape emits for a GroupNorm of `groups` groups; 0: cannot run resident */
size_t rn_mb_resident_rows(int kind, int n, int h, int wd, int cin, int cout, int stride, int groups, rn_mb_rows* layout);
int rn_mb_resident_fwd(const rn_mb_phase* phases, int nphase, int n, void* sync, rn_stream_t stream);

/* the gradient of a conv output y, as a backward kernel loads it */
typedef struct rn_mb_dy {
  const float* dy;            /* plain gradient [n,hw,c]; or NULL: computed from the fields below while loading */
  const rn_mb_norm* norm;     /* the GroupNorm block behind y (mean / rstd as the forward pass wrote them) */
  const float* g;             /* [n,hw,c] the gradient entering that block from behind */
  int32_t g_plain;            /* 0: g already carries act' and the dropout mask (a data-gradient epilogue wrote it);
                                 1: g is the gradient of the block's OUTPUT: the dropout mask is applied while loading
                                    (the block's activation must be RN_ACT_NONE) */
  rn_mb_rows grows;           /* the rows (sum gamma g, sum gamma g xhat) g's producer wrote */
} rn_mb_dy;
/* what a data-gradient kernel does with its result d = dL/d(input of the conv) */
typedef struct rn_mb_gout {
  float* out;                 /* [n,hw,c] */
  const float* add1; const float* add2;   /* optional, d += add1 + add2 (the residual path's gradient, a pyramid tap's gradient) */
  const rn_mb_norm* norm;     /* NULL: out = d.  Else the conv's input was norm's block output: g = d act'(z) mask */
  int32_t store_plain;        /* with norm: 1: out = d (its loader applies the mask: rn_mb_dy.g_plain), 0: out = g */
  rn_mb_rows grows;           /* out: rows of (sum gamma g, sum gamma g xhat) */
  float* planes;              /* out: [2][n * grows.rows_per_sample][c] per-channel (sum g | sum g xhat) of every row;
                                 rn_reduce_rows over the rows gives dbeta | dgamma */
} rn_mb_gout;
/* both gradients of rn_mb_pointwise_fwd in one launch: gout->out = dy w^T (+ ...), dw = A^T dy (A = x or the block of `in`).
 * workspace: rn_mb_pointwise_bwd_workspace bytes (weight-gradient partial sums, summed by the deferred reduction). */
size_t rn_mb_pointwise_bwd_rows(int n, int hw, int cin, int cout, int groups, rn_mb_rows* layout);   /* layout of gout->grows */
size_t rn_mb_pointwise_bwd_workspace(int n, int hw, int cin, int cout);
int rn_mb_pointwise_bwd(const float* x, const rn_mb_norm* in, const rn_mb_dy* dy, const float* w, float* dw, const rn_mb_gout* gout,
                        int n, int hw, int cin, int cout, void* workspace, size_t workspace_bytes, rn_stream_t stream,
                        rn_reduce_list* defer);
/* both gradients of rn_mb_depthwise_fwd in one launch: gout (norm = in) receives the data gradient, dw [3,3,c] */
size_t rn_mb_depthwise_bwd_rows(int n, int h, int w, int c, int stride, int groups, rn_mb_rows* layout);
size_t rn_mb_depthwise_bwd_workspace(int n, int h, int w, int c, int stride);
int rn_mb_depthwise_bwd(const rn_mb_norm* in, const rn_mb_dy* dy, const float* w, float* dw, const rn_mb_gout* gout, int n, int h,
                        int wd, int stride, void* workspace, size_t workspace_bytes, rn_stream_t stream, rn_reduce_list* defer);

/* ------------------------------------------------------------------ IoU
 * Replaces utils.iou (utils.py:62-97; known answers utils_test.py:99-118): boxes are corners [y1, x1, y2, x2].
 * pairwise != 0: out[i * nb + j] = IoU(a[i], b[j]) -- the [O,1,1,1,4] x [1,H,W,A,4] -> [O,H,W,A] broadcast of
 * dataset.py:57-60; pairwise == 0 (na == nb): out[i] = IoU(a[i], b[i]).  Boxes that do not overlap give 0
 * (utils.py:82-86); `malformed` (one int32, zeroed by the caller) is set to 1 if any box has y2 < y1 or x2 < x1
 * (the reference's tf.assert_* at utils.py:65-68).  Same float32 operation order as the assignment kernel.
 */
int rn_iou(const float* a, int64_t na, const float* b, int64_t nb, int pairwise, float* out, int32_t* malformed,
           rn_stream_t stream);

/* ------------------------------------------------------------------ anchor assignment
 * Replaces dataset.level_labels / build_labels (dataset.py:43-142) for a batch of images:
 * IoU of every anchor with every object -> arg-max/max -> one-hot class (zero where IoU<0.5),
 * log-space regression target of the arg-max object, trainable = IoU<0.4 || IoU>=0.5.
 * Degenerate objects (zero / negative extent) behave as in the reference's reduce_sum(regression * one_hot, 0)
 * (dataset.py:118-121): the log-size component of every anchor NOT assigned to such an object is NaN (-inf * 0).
 * boxes [nimg, max_obj, 4] normalised corners, class_ids [nimg, max_obj], num_obj [nimg] (>=1).
 * anchor_sizes [A,2] already divided by the image size (levels.py:38-44, dataset.py:53).
 */
int rn_anchor_assign(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg, int max_obj,
                     const float* anchor_sizes, int num_anchors, int grid_h, int grid_w, int num_classes,
                     float* cls_out, float* reg_out, uint8_t* trainable_out, int32_t* argmax_out,
                     rn_stream_t stream);
/* build_labels (dataset.py:126-142): the same for every pyramid level of the batch in one launch. */
#define RN_MAX_LEVELS 8
typedef struct rn_assign_level {
  const float* anchor_sizes; /* [A,2] of this level, normalised */
  int grid_h, grid_w;
  float* cls_out;            /* [nimg, grid_h, grid_w, A, C] */
  float* reg_out;            /* [nimg, grid_h, grid_w, A, 4] */
  uint8_t* trainable_out;    /* [nimg, grid_h, grid_w, A] */
  int32_t* argmax_out;       /* optional (NULL): index of the matched object */
} rn_assign_level;
int rn_anchor_assign_levels(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg, int max_obj,
                            const rn_assign_level* levels, int nlevel, int num_anchors, int num_classes,
                            rn_stream_t stream);
/* The reference's batch of two (dataset.py:182-204: [sample, augmentation.flip(sample)], augmentation.py:5-22) straight from
 * the assignment: every source image i is assigned ONCE and written to two batch slots -- [2i] as is, [2i+1] with every
 * map reversed along W and the x shift (regression component 1) negated -- so the mirror image's labels are the flipped
 * MAPS bit for bit (assigning mirrored boxes rounds 1 - x differently).  Outputs hold 2 * nimg images. */
int rn_anchor_assign_levels_pair(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg, int max_obj,
                                 const rn_assign_level* levels, int nlevel, int num_anchors, int num_classes,
                                 rn_stream_t stream);

/* ------------------------------------------------------------------ decode + NMS
 * rn_decode_boxes: utils.regression_postprocess (utils.py:108-117): exp, anchor scale, add
 *   cell centre, centre->corner.  reg [n,h,w,A,4] -> boxes [n,h,w,A,4].
 * rn_detect_*: utils.boxes_decode + merge_boxes_decoded + nms_classwise (utils.py:183-227,
 *   tf.image.non_max_suppression max 1000 / IoU>0.5) for a whole batch in one pass.
 */
int rn_decode_boxes(const float* reg, const float* anchor_sizes, float* boxes, int n, int h, int w, int num_anchors,
                    rn_stream_t stream);

typedef struct rn_det_level {
  const void* prob;   /* [n, rows_per_image, C] class probabilities (post-sigmoid), fp32 or fp16 (prob_f16) */
  const float* boxes; /* [n, rows_per_image, 4] decoded corner boxes, or NULL: decode on the fly from the fields below */
  int64_t rows_per_image;
  /* boxes == NULL: only the rows that become candidates are decoded (utils.regression_postprocess arithmetic,
   * utils.py:100-117, same bits as rn_decode_boxes) -- ~1 % of the rows instead of a full pass over the level */
  const void* regression;    /* [n, grid_h, grid_w, num_anchors, 4] raw network output, fp32 or fp16 (regression_f16) */
  const float* anchor_sizes; /* [num_anchors, 2] normalised (h, w)                      */
  int32_t grid_h, grid_w, num_anchors;
  /* BASELINE configs[4] (fp16 inference): the maps stay in the storage type the net wrote them in -- 2 bytes per
   * element through the scan instead of 4; all arithmetic (max, threshold, decode, IoU) is fp32 on the exact values. */
  int32_t prob_f16;        /* != 0: prob holds fp16                                                                */
  int32_t regression_f16;  /* != 0: regression holds fp16                                                           */
  /* != 0: prob holds the class LOGITS (classification.unscaled); the scan applies the sigmoid of train.py:74 /
   * utils.py:246 element by element on the fly (same expression as rn_act_fwd(RN_ACT_SIGMOID), so fp32 results
   * equal sigmoid-then-scan bit for bit) -- the probability map is never written or re-read */
  int32_t prob_is_logit;
} rn_det_level;

typedef struct rn_det_params {
  int32_t n, num_classes, max_per_class; /* 1000 = utils.NMS_MAX_OUTPUT_SIZE          */
  float score_threshold, iou_threshold;  /* 0.5 / 0.5                                  */
  int64_t max_candidates;                /* capacity of the candidate buffers          */
} rn_det_params;

size_t rn_detect_workspace(const rn_det_level* levels, int nlevels, const rn_det_params* p);
/* outputs (device): out_boxes [max_candidates,4], out_scores, out_class (int32), out_image (int32),
 * out_anchor (int64 flat index of the row inside its image, levels concatenated P3..P7):
 * survivors in (image, class, score-desc, index-asc) order; counts[0]=#candidates,
 * counts[1]=#survivors, counts[2..2+n) survivors per image. */
int rn_detect(const rn_det_level* levels, int nlevels, const rn_det_params* p, float* out_boxes, float* out_scores,
              int32_t* out_class, int32_t* out_image, int64_t* out_anchor, int64_t* counts, void* workspace,
              size_t workspace_bytes, rn_stream_t stream);

/* utils.boxes_decode alone (utils.py:183-195) for a batch: candidates (max prob > threshold) in
 * row-major anchor order, no sort / NMS.  Same outputs as rn_detect; counts[0]=counts[1]=#candidates. */
int rn_boxes_decode(const rn_det_level* levels, int nlevels, const rn_det_params* p, float* out_boxes,
                    float* out_scores, int32_t* out_class, int32_t* out_image, int64_t* out_anchor, int64_t* counts,
                    void* workspace, size_t workspace_bytes, rn_stream_t stream);
/* utils.nms_classwise / utils.nms (utils.py:198-220) on already decoded boxes: K = *count_dev
 * (device scalar) rows of boxes/scores/class_ids/image_ids; p->max_candidates >= K is the buffer
 * capacity.  Output order (image, class, score desc, index asc); out_index = row of the input. */
size_t rn_nms_classwise_workspace(const rn_det_params* p);
int rn_nms_classwise(const float* boxes, const float* scores, const int32_t* class_ids, const int32_t* image_ids,
                     const int64_t* count_dev, const rn_det_params* p, float* out_boxes, float* out_scores,
                     int32_t* out_class, int32_t* out_image, int64_t* out_index, int64_t* counts, void* workspace,
                     size_t workspace_bytes, rn_stream_t stream);

/* ------------------------------------------------------------------ optimizer
 * Replaces tf.train.{Momentum,RMSProp,Adam}Optimizer.apply + the L2 regulariser gradient
 * + tf.clip_by_global_norm (train.py:111-134,221) on ONE flat fp32 parameter arena.
 * The arena is cut into blocks of RN_OPT_BLOCK elements; wd_per_block[b] is the L2 scale of
 * the parameter that owns block b (0 for gamma/beta/bias and padding).
 *   g' = grad*grad_scale*clip_scale + wd*w      (grad_scale = 1/world_size)
 */
#define RN_OPT_BLOCK 1024
size_t rn_optimizer_workspace(int64_t count);
/* out[0] = sum(grad^2)*grad_scale^2 (global norm squared), out[1] = sum_b wd_b * 0.5*sum(w^2) */
int rn_grad_norm_l2reg(const float* w, const float* grad, const float* wd_per_block, int64_t count,
                       float grad_scale, float* out2, void* workspace, size_t workspace_bytes, rn_stream_t stream);
/* clip_norm <= 0 disables clipping; norm_sq is the device scalar from rn_grad_norm_l2reg.
 * step is 1-based (Adam bias correction). state1/state2: momentum-acc | rms,mom | m,v.
 * advance_counter (optional): a device word incremented by advance_by in the same launch -- the per-step counter the
 * fused dropouts hash (rn_gn_params.drop_seed_dev), so that the next step (or hipGraph replay) draws fresh masks
 * (tf.layers.Dropout draws new noise every session.run, mobilenet_v2.py:62). */
int rn_optimizer_step(int kind, float* w, const float* grad, float* state1, float* state2,
                      const float* wd_per_block, int64_t count, float lr, float grad_scale, float clip_norm,
                      const float* norm_sq, int64_t step, uint64_t* advance_counter, uint64_t advance_by, rn_stream_t stream);
/* The same update WITHOUT clipping, for a slice [w, w + count) of the arena, that also forms the slice's share of
 * (sum g'^2, L2 regulariser value) in the same pass (one read of w and grad instead of two): rn_optimizer_norm_pairs(count)
 * (double, double) pairs into `partial`; rn_norm_reg_finalize sums the pairs of all slices (fixed order) into out2[2] =
 * what rn_grad_norm_l2reg writes.  A trainer can so update the heads + FPN slice while the backbone's backward pass runs. */
int64_t rn_optimizer_norm_pairs(int64_t count);
int rn_optimizer_step_norm(int kind, float* w, const float* grad, float* state1, float* state2, const float* wd_per_block,
                           int64_t count, float lr, float grad_scale, int64_t step, uint64_t* advance_counter,
                           uint64_t advance_by, double* partial, rn_stream_t stream);
int rn_norm_reg_finalize(const double* partial, int64_t npairs, float* out2, rn_stream_t stream);
/* *counter += inc on the stream (the same counter, for callers that run backward passes without an optimizer step) */
int rn_counter_add(uint64_t* counter, uint64_t inc, rn_stream_t stream);
/* p[0..count) = 0 (16-byte aligned): the gradient arena before a backward pass (the reference's graph zero-initialises
 * its gradient accumulators, train.py:121-123) */
int rn_zero(float* p, int64_t count, rn_stream_t stream);

/* out = a + b for up to RN_MAX_SEG tensors in one launch (count floats each, 16-byte aligned): the sum autograd forms when
 * a tensor feeds two branches -- a bottleneck's input (expand conv + identity, mobilenet_v2.py:91-92), a pyramid level
 * (class + box subnet, retinanet.py:283-291), C5 (P5 + P6, retinanet.py:214-216). */
typedef struct rn_add_seg { const float* a; const float* b; float* out; int64_t count; } rn_add_seg;
int rn_add_segs(const rn_add_seg* segs, int nseg, rn_stream_t stream);

/* Measurement aid (bench.py `config.collective_standin`; not on the product path): what a ring all-reduce under the
 * backbone's backward pass would take from the compute stream on ONE GPU -- `blocks` workgroups of 256 threads (RCCL runs
 * one workgroup per channel) stream `bytes` from src to dst (16-byte aligned), paced by the constant 100 MHz clock so the
 * copy lasts ~target_us (2 (R-1)/R x slice bytes at the ~150 GB/s a ring gets from one xGMI link).  Launch it on a side
 * stream.  The reference's MirroredStrategy (train.py:261-267) has no counterpart: this replaces nothing. */
int rn_debug_collective_standin(const void* src, void* dst, int64_t bytes, int blocks, float target_us, rn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RN_HIP_H */
