/* Plain-C restatement of the fused hot path, forward and backward -- TEST INFRASTRUCTURE ONLY.
 *
 * Never linked into, imported or called by the product (waldo_amd/); only tests/ and
 * __graft_entry__.build() touch it.  Double precision throughout: it plays the "exact" side in the
 * parity tests next to the fp32 torch restatement (oracle/wif_oracle.py), and is itself pinned by
 * the reference's own outputs and gradients (tests/golden/warp_composite_*.npz).
 *
 * The chain, in the reference's order:
 *   TPSWarp.__init__ / forward     models/modules/warp.py:15-18, 21-55   (kernel_distance, K^-1, repr)
 *   F.grid_sample defaults         bilinear, zeros, align_corners=False  (ix = ((x+1)*W-1)/2)
 *   LVD.reduce_comp                models/nets/lvd.py:100-114            (bg alpha 1, occlusion product)
 *   pixel grid                     tools/utils.py:293-297                (x_j = -1 + (2j+1)/W)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double phi(double ax, double ay, double bx, double by) {
  /* kernel_distance, warp.py:15-18: d = |a|^2 + |b|^2 - 2 a.b ; 0.5 * d * log(d + 1e-8) */
  const double d = ax * ax + ay * ay + bx * bx + by * by - 2.0 * (ax * bx + ay * by);
  return 0.5 * d * log(d + 1e-8);
}

/* in-place Gauss-Jordan inverse with partial pivoting; returns 0 on success */
static int invert(double* a, int n) {
  double* inv = (double*)calloc((size_t)n * n, sizeof(double));
  if (!inv) return -1;
  for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(a[r * n + c]) > fabs(a[piv * n + c])) piv = r;
    if (fabs(a[piv * n + c]) < 1e-300) {
      free(inv);
      return -2;
    }
    if (piv != c)
      for (int j = 0; j < n; ++j) {
        double t = a[c * n + j];
        a[c * n + j] = a[piv * n + j];
        a[piv * n + j] = t;
        t = inv[c * n + j];
        inv[c * n + j] = inv[piv * n + j];
        inv[piv * n + j] = t;
      }
    const double d = 1.0 / a[c * n + c];
    for (int j = 0; j < n; ++j) {
      a[c * n + j] *= d;
      inv[c * n + j] *= d;
    }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double m = a[r * n + c];
      if (m == 0.0) continue;
      for (int j = 0; j < n; ++j) {
        a[r * n + j] -= m * a[c * n + j];
        inv[r * n + j] -= m * inv[c * n + j];
      }
    }
  }
  memcpy(a, inv, (size_t)n * n * sizeof(double));
  free(inv);
  return 0;
}

typedef struct {
  int x0, y0;
  double fx, fy;
  int in[4];   /* corner (y0,x0) (y0,x1) (y1,x0) (y1,x1) inside the image */
  int idx[4];  /* texel index of the corner (valid when in[] is set) */
  double w[4];
} taps_t;

static void make_taps(double gx, double gy, int H, int W, taps_t* t) {
  const double ix = ((gx + 1.0) * W - 1.0) / 2.0, iy = ((gy + 1.0) * H - 1.0) / 2.0;
  const double x0f = floor(ix), y0f = floor(iy);
  t->fx = ix - x0f;
  t->fy = iy - y0f;
  /* far-away coordinates: all corners outside; keep the int conversion defined */
  t->x0 = (x0f < -4.0 || x0f > W + 4.0) ? -4 : (int)x0f;
  t->y0 = (y0f < -4.0 || y0f > H + 4.0) ? -4 : (int)y0f;
  if (!(ix == ix) || !(iy == iy)) t->x0 = t->y0 = -4; /* NaN */
  const double wx[2] = {1.0 - t->fx, t->fx}, wy[2] = {1.0 - t->fy, t->fy};
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int x = t->x0 + dx, y = t->y0 + dy, k = dy * 2 + dx;
      t->in[k] = x >= 0 && x < W && y >= 0 && y < H;
      t->idx[k] = t->in[k] ? y * W + x : 0;
      t->w[k] = wx[dx] * wy[dy];
    }
}

/* loss: weights mode  sum(rgb * w_rgb) + sum(alpha * w_alpha)   (either may be NULL)
 *       square mode   mean(rgb^2)                                (loss_sq != 0)
 * delta: every layer is sampled as grid_sample(x + delta) - delta (Warper.obj_to_output /
 *       bg_to_output, lvd.py:548,559); 0 is plain F.grid_sample.
 * outputs (all double, caller-allocated, grad_* may be NULL to skip the backward):
 *   rgb (F,3,HW)  alpha (F,L,HW)  grad_layers (F,L,4,HW)  grad_pts (F*L,N,2)  grad_occ (F,L,L) */
int waldo_oracle_fused(const float* layers, const float* pts, const float* occ, const float* ctrl,
                       int F, int L, int H, int W, int N, const float* w_rgb, const float* w_alpha,
                       int loss_sq, double delta, double* rgb, double* alpha, double* grad_layers,
                       double* grad_pts, double* grad_occ) {
  const int K3 = N + 3, HW = H * W;
  if (F < 0 || L < 1 || L > 64 || H < 1 || W < 1 || N < 1 || N > 64) return -1;
  double* kinv = (double*)calloc((size_t)K3 * K3, sizeof(double));
  double* repr = (double*)malloc((size_t)HW * K3 * sizeof(double));
  double* mapping = (double*)malloc((size_t)K3 * 2 * sizeof(double));
  double* grid = (double*)malloc((size_t)L * HW * 2 * sizeof(double));
  double* ggrid = (double*)malloc((size_t)L * HW * 2 * sizeof(double));
  if (!kinv || !repr || !mapping || !grid || !ggrid) return -2;
  /* TPSWarp.__init__ (warp.py:24-46) */
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; ++j) kinv[i * K3 + j] = phi(ctrl[2 * i], ctrl[2 * i + 1], ctrl[2 * j], ctrl[2 * j + 1]);
    kinv[i * K3 + N] = kinv[N * K3 + i] = 1.0;
    kinv[i * K3 + N + 1] = kinv[(N + 1) * K3 + i] = ctrl[2 * i];
    kinv[i * K3 + N + 2] = kinv[(N + 2) * K3 + i] = ctrl[2 * i + 1];
  }
  if (invert(kinv, K3)) return -3;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const double gx = -1.0 + (2.0 * x + 1.0) / W, gy = -1.0 + (2.0 * y + 1.0) / H;
      double* r = repr + (size_t)(y * W + x) * K3;
      for (int k = 0; k < N; ++k) r[k] = phi(gx, gy, ctrl[2 * k], ctrl[2 * k + 1]);
      r[N] = 1.0;
      r[N + 1] = gx;
      r[N + 2] = gy;
    }
  if (grad_layers) memset(grad_layers, 0, sizeof(double) * (size_t)F * L * 4 * HW);
  if (grad_pts) memset(grad_pts, 0, sizeof(double) * (size_t)F * L * N * 2);
  if (grad_occ) memset(grad_occ, 0, sizeof(double) * (size_t)F * L * L);
  const int backward = grad_layers || grad_pts || grad_occ;
  const double nrgb = (double)F * 3 * HW;

  for (int f = 0; f < F; ++f) {
    /* TPSWarp.forward (warp.py:49-55) for the L layers of this frame */
    for (int l = 0; l < L; ++l) {
      const float* p = pts + (size_t)(f * L + l) * N * 2;
      for (int k = 0; k < K3; ++k)
        for (int d = 0; d < 2; ++d) {
          double s = 0.0;
          for (int n = 0; n < N; ++n) s += kinv[k * K3 + n] * p[n * 2 + d];
          mapping[k * 2 + d] = s;
        }
      for (int q = 0; q < HW; ++q)
        for (int d = 0; d < 2; ++d) {
          double s = 0.0;
          for (int k = 0; k < K3; ++k) s += repr[(size_t)q * K3 + k] * mapping[k * 2 + d];
          grid[((size_t)l * HW + q) * 2 + d] = s;
        }
    }
    memset(ggrid, 0, sizeof(double) * (size_t)L * HW * 2);
    const float* oc = occ + (size_t)f * L * L;
    for (int q = 0; q < HW; ++q) {
      double v[64][4], a[64], ap[64];
      taps_t tp[64];
      for (int l = 0; l < L; ++l) {
        make_taps(grid[((size_t)l * HW + q) * 2], grid[((size_t)l * HW + q) * 2 + 1], H, W, &tp[l]);
        for (int c = 0; c < 4; ++c) {
          const float* plane = layers + ((size_t)(f * L + l) * 4 + c) * HW;
          double s = 0.0;
          for (int k = 0; k < 4; ++k)
            if (tp[l].in[k]) s += tp[l].w[k] * (plane[tp[l].idx[k]] + delta);
          s -= delta;
          v[l][c] = (s + 1.0) / 2.0; /* reduce_comp, lvd.py:103 */
        }
        a[l] = l == 0 ? 1.0 : v[l][3]; /* lvd.py:105 */
      }
      double out[3] = {0.0, 0.0, 0.0};
      for (int j = 0; j < L; ++j) {
        double pr = 1.0;
        for (int i = 0; i < L; ++i) pr *= 1.0 - a[i] * oc[i * L + j]; /* lvd.py:109-110 */
        ap[j] = a[j] * pr;
        for (int c = 0; c < 3; ++c) out[c] += ap[j] * v[j][c]; /* lvd.py:111 */
        alpha[((size_t)f * L + j) * HW + q] = 2.0 * ap[j] - 1.0;
      }
      for (int c = 0; c < 3; ++c) rgb[((size_t)f * 3 + c) * HW + q] = 2.0 * out[c] - 1.0;
      if (!backward) continue;
      /* ---- backward at this pixel */
      double go[3], gap[64], ga[64];
      for (int c = 0; c < 3; ++c) {
        const size_t o = ((size_t)f * 3 + c) * HW + q;
        go[c] = loss_sq ? 2.0 * (2.0 * out[c] - 1.0) / nrgb : (w_rgb ? (double)w_rgb[o] : 0.0);
      }
      for (int j = 0; j < L; ++j) {
        gap[j] = (!loss_sq && w_alpha) ? 2.0 * w_alpha[((size_t)f * L + j) * HW + q] : 0.0;
        for (int c = 0; c < 3; ++c) gap[j] += 2.0 * go[c] * v[j][c];
        ga[j] = 0.0;
      }
      for (int j = 0; j < L; ++j) {
        double pr = 1.0;
        for (int i = 0; i < L; ++i) pr *= 1.0 - a[i] * oc[i * L + j];
        ga[j] += gap[j] * pr;
        for (int m = 0; m < L; ++m) {
          double ex = 1.0; /* product over i != m */
          for (int i = 0; i < L; ++i)
            if (i != m) ex *= 1.0 - a[i] * oc[i * L + j];
          ga[m] += gap[j] * a[j] * (-(double)oc[m * L + j]) * ex;
          if (grad_occ) grad_occ[(size_t)f * L * L + m * L + j] += gap[j] * a[j] * (-a[m]) * ex;
        }
      }
      for (int l = 0; l < L; ++l) {
        double gs[4]; /* d loss / d sample (before the (s+1)/2) */
        for (int c = 0; c < 3; ++c) gs[c] = 0.5 * 2.0 * go[c] * ap[l];
        gs[3] = l == 0 ? 0.0 : 0.5 * ga[l];
        double gix = 0.0, giy = 0.0;
        const taps_t* t = &tp[l];
        for (int c = 0; c < 4; ++c) {
          const float* plane = layers + ((size_t)(f * L + l) * 4 + c) * HW;
          double tex[4];
          for (int k = 0; k < 4; ++k) {
            tex[k] = t->in[k] ? (double)plane[t->idx[k]] + delta : 0.0;
            if (t->in[k] && grad_layers) grad_layers[((size_t)(f * L + l) * 4 + c) * HW + t->idx[k]] += gs[c] * t->w[k];
          }
          /* d s / d ix, d s / d iy of the bilinear interpolant (zero-padded texels) */
          gix += gs[c] * ((1.0 - t->fy) * (tex[1] - tex[0]) + t->fy * (tex[3] - tex[2]));
          giy += gs[c] * ((1.0 - t->fx) * (tex[2] - tex[0]) + t->fx * (tex[3] - tex[1]));
        }
        ggrid[((size_t)l * HW + q) * 2] = gix * W / 2.0;
        ggrid[((size_t)l * HW + q) * 2 + 1] = giy * H / 2.0;
      }
    }
    if (grad_pts) {
      for (int l = 0; l < L; ++l) {
        double gm[67 * 2];
        for (int k = 0; k < K3 && k < 67; ++k)
          for (int d = 0; d < 2; ++d) {
            double s = 0.0;
            for (int q = 0; q < HW; ++q) s += repr[(size_t)q * K3 + k] * ggrid[((size_t)l * HW + q) * 2 + d];
            gm[k * 2 + d] = s;
          }
        for (int n = 0; n < N; ++n)
          for (int d = 0; d < 2; ++d) {
            double s = 0.0;
            for (int k = 0; k < K3; ++k) s += kinv[k * K3 + n] * gm[k * 2 + d];
            grad_pts[((size_t)(f * L + l) * N + n) * 2 + d] = s;
          }
      }
    }
  }
  free(kinv);
  free(repr);
  free(mapping);
  free(grid);
  free(ggrid);
  return 0;
}
