/* oracle/c/conv_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into libfdt_hip.so, never on the product path).
 *
 * A plain-C, loop-nest restatement of the floating-point building blocks the PyTorch-based oracle
 * (oracle/pyramidbox.py) takes from ATen: nn.Conv2d (stride / padding / dilation / groups, optional bias),
 * eval-mode BatchNorm2d, ReLU / ReLU6, max_pool2d(3, s, 1) and bilinear x2 upsampling with align_corners=False.
 * It exists to pin that third-party arithmetic independently: tests/test_oracle_c.py checks that the torch
 * functions the oracle calls agree with these loops (products accumulated in double, one rounding to float, so the
 * result is the correctly rounded value up to 1 ulp regardless of summation order).
 *
 * Reference call sites restated: pyramid.py:14 (Conv2d), :98-102 (conv-bn-relu), :230 (max_pool2d), :65
 * (F.upsample bilinear), pyramid_mb2_try3.py:96,113 (depthwise Conv2d, groups == channels).
 */
#include <math.h>
#include <stddef.h>

void oracle_conv2d(const float* x, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, int K,
                   int stride, int pad, int dil, int groups, float* y) {
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int cig = Cin / groups, cog = Cout / groups;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      const int g = co / cog;
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = bias ? (double)bias[co] : 0.0;
          for (int ci = 0; ci < cig; ++ci)
            for (int ky = 0; ky < K; ++ky) {
              const int iy = oy * stride - pad + ky * dil;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * stride - pad + kx * dil;
                if (ix < 0 || ix >= W) continue;
                acc += (double)x[(((size_t)b * Cin + g * cig + ci) * H + iy) * W + ix] *
                       (double)w[(((size_t)co * cig + ci) * K + ky) * K + kx];
              }
            }
          y[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = (float)acc;
        }
    }
}

/* eval-mode BatchNorm2d: (x - mean) / sqrt(var + eps) * gamma + beta, then act (0 none, 1 ReLU, 2 ReLU6) */
void oracle_bn_act(float* x, int B, int C, int HW, const float* gamma, const float* beta, const float* mean,
                   const float* var, double eps, int act) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const double s = (double)gamma[c] / sqrt((double)var[c] + eps);
      float* p = x + ((size_t)b * C + c) * HW;
      for (int i = 0; i < HW; ++i) {
        double v = ((double)p[i] - (double)mean[c]) * s + (double)beta[c];
        if (act >= 1 && v < 0.0) v = 0.0;
        if (act == 2 && v > 6.0) v = 6.0;
        p[i] = (float)v;
      }
    }
}

void oracle_maxpool3(const float* x, int BC, int H, int W, int stride, float* y) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  for (int n = 0; n < BC; ++n)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox) {
        float m = -INFINITY;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int iy = oy * stride + dy, ix = ox * stride + dx;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float v = x[((size_t)n * H + iy) * W + ix];
            if (v > m) m = v;
          }
        y[((size_t)n * Ho + oy) * Wo + ox] = m;
      }
}

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) / 2 - 0.5, clamped at 0 */
void oracle_upsample2x(const float* x, int BC, int H, int W, float* y) {
  const int Ho = 2 * H, Wo = 2 * W;
  for (int n = 0; n < BC; ++n)
    for (int oy = 0; oy < Ho; ++oy) {
      float sy = 0.5f * ((float)oy + 0.5f) - 0.5f;
      if (sy < 0.f) sy = 0.f;
      int y0 = (int)sy;
      if (y0 > H - 1) y0 = H - 1;
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const float ly = sy - (float)y0;
      for (int ox = 0; ox < Wo; ++ox) {
        float sx = 0.5f * ((float)ox + 0.5f) - 0.5f;
        if (sx < 0.f) sx = 0.f;
        int x0 = (int)sx;
        if (x0 > W - 1) x0 = W - 1;
        const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float lx = sx - (float)x0;
        const float* p = x + (size_t)n * H * W;
        const double top = (1.0 - lx) * p[y0 * W + x0] + (double)lx * p[y0 * W + x1];
        const double bot = (1.0 - lx) * p[y1 * W + x0] + (double)lx * p[y1 * W + x1];
        y[((size_t)n * Ho + oy) * Wo + ox] = (float)((1.0 - ly) * top + (double)ly * bot);
      }
    }
}
