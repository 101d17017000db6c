// Small HBM-bound kernels around the convolutions: frame ingest, pooling, depthwise 3x3, head
// finalisation (max-in-out + 2-way softmax + NHWC flatten).  Device pointers, async on `st`.
#pragma once
#include <hip/hip_runtime.h>

namespace fdt {

// u8 BGR HWC -> f32 NCHW, minus per-channel mean (iouTracke_cal.py:41-44) then / scale
// (FaceBox: mean 0, divide by 255: FACEBOX/My_test_facebox.py:14-15).
int launch_preprocess(const unsigned char* frames, int B, int H, int W, float m0, float m1, float m2,
                      float scale, float* out, hipStream_t st);

// cv2.resize(frame, (W, H)) [INTER_LINEAR, 8UC3] fused with the ingest above: iouTracke_cal.py:123 +
// :40-46, FACEBOX/My_test_facebox.py:13-15.  frames: [B][SH][SW][3] u8.
int launch_resize_u8(const unsigned char* frames, int B, int SH, int SW, int H, int W, unsigned char* out, hipStream_t st);
int launch_resize_preprocess(const unsigned char* frames, int B, int SH, int SW, int H, int W, float m0,
                             float m1, float m2, float div, float* out, hipStream_t st);

// F.max_pool2d(x, 3, stride, 1)  (pyramid.py:230 stride 2; FACEBOX/networks.py:46 stride 1).
// relu_in: apply ReLU to the input on the fly.  crelu: input has C channels, output 2C =
// maxpool(relu(cat[x,-x])) (FACEBOX/networks.py:92-98).
int launch_maxpool3(const float* in, int B, int C, int H, int W, int stride, int crelu, float* out,
                    int Ho, int Wo, hipStream_t st);

// depthwise 3x3, pad 1, stride 1|2, + bias (folded BN) + ReLU6 (pyramid_mb2_try3.py:89-91,112-114)
int launch_dwconv3(const float* in, const float* w9, const float* bias, int B, int C, int H, int W,
                   int stride, int act, float* out, int Ho, int Wo, hipStream_t st);

// One detection level: raw head map [B][8][H][W] (ch 0-3 loc, 4-7 conf) -> loc [B][P][4] and
// softmaxed conf [B][P][2] rows [p_off, p_off+H*W): max-in-out (pyramid.py:291-305), NHWC flatten
// (:298,:306), nn.Softmax(dim=-1) (:332).
int launch_dwconv(const float* in, const float* wk, const float* bias, int B, int C, int H, int W, int K,
                  int stride, int pad, int dil, int act, float* out, int Ho, int Wo, hipStream_t st);
int launch_pad1(const float* in, int BC, int H, int W, float* out, hipStream_t st);
int launch_head_finalize(const float* head, int B, int H, int W, int level0, int P, int p_off,
                         float* loc, float* conf, float* logits, hipStream_t st);

// All detection levels of a frame in ONE launch.  A level's head map is either the finished conv output [B][8][HW]
// (ksplit == 1) or the split-K slabs the head conv left in its own workspace [B][ksplit][8][HW] (launch_conv with
// ConvArgs.defer_reduce): the slabs are summed here in ks order and the bias added last -- the order of
// splitk_reduce_kernel, so the two paths agree bit for bit -- which saves the head convs' reduce passes as well.
struct HeadLevel {
  const float* src;
  const float* bias;     // [8], used when ksplit > 1
  int HW, ksplit, level0, p_off, blk0;   // blk0: first block of this level in the grid's x dimension
  int anchors;           // 0: PyramidBox head (8 channels, max-in-out); A >= 1: FaceBoxes multibox level, channels [A*4 loc | A*2 conf]
};
struct HeadFinArgs {
  HeadLevel lv[8];
  int nlev, nblocks, P;
  float* loc;
  float* conf;
  float* logits;
};
int launch_head_finalize_all(const HeadFinArgs& a, int B, hipStream_t st);   // all levels PyramidBox heads or all multibox levels

// FaceBox multibox level: loc map [A*4][H][W] and conf map [A*2][H][W] inside one per-image block of
// `img_stride` floats -> rows of A anchors per cell (FACEBOX/multibox_layer.py:34-48), raw logits and
// softmax (FACEBOX/My_test_facebox.py:25).
int launch_multibox_finalize(const float* locmap, const float* confmap, long long img_stride, int B, int A,
                             int H, int W, int P, int p_off, float* loc, float* conf, float* logits,
                             hipStream_t st);

// Fused conv[0..5] of an InvertedResidual with expand_ratio != 1 (pyramid_mb2_try3.py:96-114): 1x1 expand + BN + ReLU6
// + depthwise 3x3 (stride 1|2, pad 1) + BN + ReLU6; BatchNorms folded into (w1 [hid][Cin], b1) and (wdw [hid][9], bdw).
// The expanded tensor stays in LDS (fused_ir.hip).
int launch_expand_dw(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                     const float* bdw, int hid, int stride, float* out, int Ho, int Wo, hipStream_t st, int device = -1);
size_t expand_dw_lds_bytes(int Cin, int stride, int hid);
// ... and with the 1x1 project conv + BN (+ residual) of the block in the same kernel (oup <= 32): the whole InvertedResidual
int launch_ir_block(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                    const float* bdw, int hid, int stride, const float* wp, const float* bp, int oup, int residual, float* out,
                    int Ho, int Wo, hipStream_t st, int dev);
size_t ir_block_lds_bytes(int Cin, int stride, int hid);

// Streaming vector-ALU kernels of the MobileNetV2 detectors' HBM-bound front (stream_ir.hip).
// depthwise 3x3 (stride 1, pad 1) + bias + ReLU6, then the 1x1 project + bias (+ residual [B][oup][H][W]) in one pass: in
// [B][hid][H][W] -> out [B][oup][H][W]; w9 [hid][9], bdw [hid], wpt [hid][oup] (transposed), bp [oup]; oup in {16, 24, 32}.
bool dw_project_supported(int hid, int H, int W, int oup);
int launch_dw_project(const float* in, int B, int hid, int H, int W, const float* w9, const float* bdw, const float* wpt,
                      const float* bp, int oup, const float* res, float* out, hipStream_t st);
// Conv2d(3, 32, 3, stride 2, pad 1) + bias + act on raw uint8 HWC BGR frames ((float)u8 - mean in registers):
// frames [B][H][W][3] -> out [B][32][Ho][Wo]; wt [27][32] ((c, dy, dx) major); W % 8 == 0.
bool stem3x3s2_u8_supported(int H, int W, int Cout);
int launch_stem3x3s2_u8(const unsigned char* frames, int B, int H, int W, const float mean[3], const float* wt, const float* bias,
                        int Cout, int act, float* out, hipStream_t st);

}  // namespace fdt
