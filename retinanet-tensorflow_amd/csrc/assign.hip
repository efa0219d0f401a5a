// Anchor assignment on the device (dataset.level_labels / build_labels, dataset.py:43-142).
// Compiled with -ffp-contract=off: every float32 operation is rounded separately, in the
// reference's order, so the class maps, trainable masks and arg-max indices are bit-identical
// to the CPU oracle (BASELINE north_star: "bit-exact for anchor assignment").
//
// One thread per anchor (image, cell, anchor).  The [.., C] one-hot rows dominate the bytes
// (C floats per anchor), so a wave writes the rows of its 64 anchors cooperatively: lane l
// stores class l (and l+64..) of anchor i for i = 0..63 -- contiguous 4-B stores.
#include "rn_common.h"

namespace {
constexpr int T = 256;

struct LevelArgs {
  const float* anchor_sizes; int H, W;
  float* cls_out; float* reg_out; uint8_t* tr_out; int32_t* arg_out;
  int64_t iter0;  // first 64-anchor wave iteration of this level (levels laid end to end)
};
struct AssignArgs {
  const float* boxes; const int32_t* class_ids; const int32_t* num_obj;
  int nimg, max_obj, A, C, nlevel;
  int pair;       // 1: every source image fills TWO batch slots, [2i] = the image, [2i+1] = its horizontal mirror image
                  // (augmentation.flip of the finished maps: W axis reversed, x shift negated; dataset.py:182-204)
  int64_t iters;  // wave iterations of all levels
  LevelArgs lv[RN_MAX_LEVELS];
};

// cell centre i of `size` cells: tf.linspace(cell/2, 1-cell/2, size)[i] in float32
// (dataset.py:16-25; [TF-sem] value = start + step*i, step = (stop-start)/(size-1))
__device__ __forceinline__ float cell_center(int i, int size) {
  const float cell = (float)(1.0 / (double)size);
  const float start = cell / 2.0f;
  if (size == 1) return start;
  const float stop = 1.0f - start;
  const float step = (stop - start) / (float)(size - 1);
  const float t = step * (float)i;
  return start + t;
}

// utils.iou (utils.py:62-97) of two corner boxes, every float32 operation rounded on its own and in the reference's order
__device__ __forceinline__ float iou_corners(float a0, float a1, float a2, float a3, float area_a,
                                             float b0, float b1, float b2, float b3) {
  const float yt = fmaxf(a0, b0), xl = fmaxf(a1, b1), yb = fminf(a2, b2), xr = fminf(a3, b3);
  const bool invalid = (yb < yt) || (xr < xl);
  const float inter = (yb - yt) * (xr - xl);
  const float area_b = (b2 - b0) * (b3 - b1);
  const float den = (area_a + area_b) - inter;
  const float v = inter / den;
  return invalid ? 0.f : v;
}

__global__ __launch_bounds__(T) void iou_kernel(const float4* __restrict__ a, int64_t na, const float4* __restrict__ b,
                                                int64_t nb, int pairwise, float* __restrict__ out, int32_t* malformed) {
  const int64_t total = pairwise ? na * nb : na;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < total; i += (int64_t)gridDim.x * T) {
    const float4 pa = a[pairwise ? i / nb : i];
    const float4 pb = b[pairwise ? i % nb : i];
    if (malformed && (pa.z < pa.x || pa.w < pa.y || pb.z < pb.x || pb.w < pb.y)) *malformed = 1;
    out[i] = iou_corners(pa.x, pa.y, pa.z, pa.w, (pa.z - pa.x) * (pa.w - pa.y), pb.x, pb.y, pb.z, pb.w);
  }
}

__global__ __launch_bounds__(T) void assign_kernel(const AssignArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * T + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * T) >> 6;
  for (int64_t gi = wave0; gi < a.iters; gi += nwaves) {
    int l = 0;  // the pyramid level this wave iteration belongs to (wave-uniform)
    while (l + 1 < a.nlevel && gi >= a.lv[l + 1].iter0) ++l;
    const LevelArgs& L = a.lv[l];
    const int64_t wi = gi - L.iter0;
    const int64_t per_img = (int64_t)L.H * L.W * a.A;
    const int64_t total = per_img * a.nimg;
    const int64_t r = wi * 64 + lane;
    const bool active = r < total;
    int cls = -1;
    bool bg = true;
    int row0 = -1, row1 = -1;  // destination rows (anchors) of this lane: the image's slot and (pair mode) the mirror image's
    if (active) {
      const int img = (int)(r / per_img);
      int64_t q = r - (int64_t)img * per_img;
      const int64_t q_in_img = q;
      const int an = (int)(q % a.A); q /= a.A;
      const int x_ = (int)(q % L.W);
      const int y_ = (int)(q / L.W);
      if (a.pair) {
        row0 = (int)((int64_t)(2 * img) * per_img + q_in_img);
        row1 = (int)((int64_t)(2 * img + 1) * per_img + ((int64_t)y_ * L.W + (L.W - 1 - x_)) * a.A + an);
      } else {
        row0 = (int)r;
      }
      const float ay = cell_center(y_, L.H), ax = cell_center(x_, L.W);
      const float ah = L.anchor_sizes[an * 2], aw = L.anchor_sizes[an * 2 + 1];
      // from_center_box(anchor) (dataset.py:35-39)
      const float hh = ah / 2.0f, hw = aw / 2.0f;
      const float a0 = ay - hh, a1 = ax - hw, a2 = ay + hh, a3 = ax + hw;
      const float area_a = (a2 - a0) * (a3 - a1);
      const int no = a.num_obj[img];
      float best = 0.f;
      int best_i = 0;
      float bcy = 0.f, bcx = 0.f, bsh = 1.f, bsw = 1.f;
      // dataset.py:118-121 selects the arg-max object's target with reduce_sum(regression * one_hot, 0): an object whose
      // log-size target is not finite (zero / negative extent: log <= 0 -> -inf / NaN) turns that component into NaN for
      // every anchor it is NOT assigned to (-inf * 0).  Count such objects so the same values come out.
      int bad_h = 0, bad_w = 0;
      bool best_bad_h = false, best_bad_w = false;
      for (int o = 0; o < no; ++o) {
        const float* tb = a.boxes + ((size_t)img * a.max_obj + o) * 4;
        // to_center_box (dataset.py:28-32) then from_center_box
        const float sh = tb[2] - tb[0], sw = tb[3] - tb[1];
        const float cy = tb[0] + sh / 2.0f, cx = tb[1] + sw / 2.0f;
        const float th = sh / 2.0f, tw = sw / 2.0f;
        const float b0 = cy - th, b1 = cx - tw, b2 = cy + th, b3 = cx + tw;
        const float v = iou_corners(a0, a1, a2, a3, area_a, b0, b1, b2, b3);   // utils.iou (utils.py:62-97)
        const bool oh_bad = !(sh > 0.f), ow_bad = !(sw > 0.f);
        bad_h += oh_bad ? 1 : 0; bad_w += ow_bad ? 1 : 0;
        if (o == 0 || v > best) {  // arg-max keeps the FIRST maximum
          best = v; best_i = o; bcy = cy; bcx = cx; bsh = sh; bsw = sw;
          best_bad_h = oh_bad; best_bad_w = ow_bad;
        }
      }
      bg = best < 0.5f;                                   // dataset.py:83
      const bool trainable = (best < 0.4f) || (best >= 0.5f);  // dataset.py:87
      cls = a.class_ids[(size_t)img * a.max_obj + best_i];
      // regression target of the arg-max object, NOT zeroed on background (dataset.py:105-121)
      float4 rg;
      rg.x = (bcy - ay) / ah;
      rg.y = (bcx - ax) / aw;
      rg.z = logf(bsh / ah);
      rg.w = logf(bsw / aw);
      if (bad_h - (best_bad_h ? 1 : 0) > 0) rg.z = __builtin_nanf("");
      if (bad_w - (best_bad_w ? 1 : 0) > 0) rg.w = __builtin_nanf("");
      *reinterpret_cast<float4*>(L.reg_out + (size_t)row0 * 4) = rg;
      L.tr_out[row0] = trainable ? 1 : 0;
      if (L.arg_out) L.arg_out[row0] = best_i;
      if (a.pair) {  // augmentation.py:5-22: the maps reversed along W, the x shift negated
        rg.y = -rg.y;
        *reinterpret_cast<float4*>(L.reg_out + (size_t)row1 * 4) = rg;
        L.tr_out[row1] = trainable ? 1 : 0;
        if (L.arg_out) L.arg_out[row1] = best_i;
      }
    }
    const int hot = (active && !bg && cls >= 0 && cls < a.C) ? cls : -1;
    // cooperative one-hot rows
    const int64_t rbase = wi * 64;
    const int nrow = (int)((total - rbase) < 64 ? (total - rbase) : 64);
    for (int i = 0; i < nrow; ++i) {
      const int h = __shfl(hot, i, 64);
      float* row = L.cls_out + (size_t)__shfl(row0, i, 64) * a.C;
      for (int c = lane; c < a.C; c += 64) row[c] = (c == h) ? 1.f : 0.f;
      if (a.pair) {
        float* mrow = L.cls_out + (size_t)__shfl(row1, i, 64) * a.C;
        for (int c = lane; c < a.C; c += 64) mrow[c] = (c == h) ? 1.f : 0.f;
      }
    }
  }
}
}  // namespace

extern "C" int rn_iou(const float* a, int64_t na, const float* b, int64_t nb, int pairwise, float* out, int32_t* malformed,
                      rn_stream_t stream) {
  RN_CHECK_ARG(a && b && out, "iou: null pointer");
  RN_CHECK_ARG(na >= 0 && nb >= 0 && (pairwise || na == nb), "iou: element-wise mode needs na == nb");
  const int64_t total = pairwise ? na * nb : na;
  if (total == 0) return RN_OK;
  int64_t blocks = (total + T - 1) / T;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(iou_kernel, dim3((unsigned)blocks), dim3(T), 0, (hipStream_t)stream, (const float4*)a, na,
                     (const float4*)b, nb, pairwise, out, malformed);
  RN_LAUNCH_CHECK();
  return RN_OK;
}

namespace {
int assign_levels_impl(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg, int max_obj,
                       const rn_assign_level* levels, int nlevel, int num_anchors, int num_classes, int pair, rn_stream_t stream) {
  RN_CHECK_ARG(boxes && class_ids && num_obj && levels, "anchor_assign: null pointer");
  RN_CHECK_ARG(nlevel >= 1 && nlevel <= RN_MAX_LEVELS, "anchor_assign: 1..%d levels", RN_MAX_LEVELS);
  RN_CHECK_ARG(nimg >= 1 && max_obj >= 1 && num_anchors >= 1 && num_classes >= 1, "anchor_assign: bad shape");
  AssignArgs a = {};
  a.boxes = boxes; a.class_ids = class_ids; a.num_obj = num_obj;
  a.nimg = nimg; a.max_obj = max_obj; a.A = num_anchors; a.C = num_classes; a.nlevel = nlevel; a.pair = pair ? 1 : 0;
  int64_t iters = 0;
  for (int l = 0; l < nlevel; ++l) {
    const rn_assign_level& v = levels[l];
    RN_CHECK_ARG(v.anchor_sizes && v.cls_out && v.reg_out && v.trainable_out, "anchor_assign: null pointer (level %d)", l);
    RN_CHECK_ARG(v.grid_h >= 1 && v.grid_w >= 1, "anchor_assign: bad grid (level %d)", l);
    a.lv[l] = {v.anchor_sizes, v.grid_h, v.grid_w, v.cls_out, v.reg_out, v.trainable_out, v.argmax_out, iters};
    RN_UNSUPPORTED((double)(pair ? 2 : 1) * nimg * v.grid_h * v.grid_w * num_anchors >= 2147483648.0, "anchor_assign: level %d has >= 2^31 anchors", l);
    iters += ((int64_t)nimg * v.grid_h * v.grid_w * num_anchors + 63) / 64;
  }
  a.iters = iters;
  int64_t blocks = (iters * 64 + T - 1) / T;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(assign_kernel, dim3((unsigned)blocks), dim3(T), 0, (hipStream_t)stream, a);
  RN_LAUNCH_CHECK();
  return RN_OK;
}
}  // namespace

extern "C" int rn_anchor_assign_levels(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg,
                                       int max_obj, const rn_assign_level* levels, int nlevel, int num_anchors,
                                       int num_classes, rn_stream_t stream) {
  return assign_levels_impl(boxes, class_ids, num_obj, nimg, max_obj, levels, nlevel, num_anchors, num_classes, 0, stream);
}

extern "C" int rn_anchor_assign_levels_pair(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg,
                                            int max_obj, const rn_assign_level* levels, int nlevel, int num_anchors,
                                            int num_classes, rn_stream_t stream) {
  return assign_levels_impl(boxes, class_ids, num_obj, nimg, max_obj, levels, nlevel, num_anchors, num_classes, 1, stream);
}

extern "C" int rn_anchor_assign(const float* boxes, const int32_t* class_ids, const int32_t* num_obj, int nimg,
                                int max_obj, const float* anchor_sizes, int num_anchors, int grid_h, int grid_w,
                                int num_classes, float* cls_out, float* reg_out, uint8_t* trainable_out,
                                int32_t* argmax_out, rn_stream_t stream) {
  const rn_assign_level one = {anchor_sizes, grid_h, grid_w, cls_out, reg_out, trainable_out, argmax_out};
  return rn_anchor_assign_levels(boxes, class_ids, num_obj, nimg, max_obj, &one, 1, num_anchors, num_classes, stream);
}
