// A3: grid inversion by forward splat + hole filling -- replaces InverseWarp.forward
// (models/modules/warp.py:71-174, pad == True; any odd kernel_size -- the one-launch kernels are written for the 3 x 3
// every script uses, other sizes run pass by pass) and its autograd.
//
//   1. displacement d = src_grid - identity, bilinearly resized to the target raster
//      (F.interpolate, align_corners=False)                                       warp.py:75-79
//   2. every resized sample s lands on cell round-half-even(position + d); among several samples
//      on one cell the LOWEST sample index wins (the reference keeps the first of each run after a
//      sort; which one that is depends on the sort's stability -- the build fixes it to the
//      stable-sort answer, see oracle/ref_import.py:stable_sort)                  warp.py:80-123
//      -> integer atomicMin on the sample index, no sort
//   3. field = -d of the winner, padded by niter+1; niter Jacobi passes: the 4-neighbour ring of
//      the filled set takes the Gaussian-weighted mean of its filled 3x3 neighbours warp.py:125-151
//   4. optional erosion of the mask, niter passes                                 warp.py:153-162
//   5. unfilled cells sample far out of range (offset 2W, 2H px); crop            warp.py:164-174
// The integer path (cells, winners, masks) carries no gradient; the value path is linear and is
// differentiated exactly by replaying the passes in reverse with the stored fill order.
#include <stdlib.h>

#include "waldo_common.hip.h"

namespace waldo {

constexpr int kNoWinner = 0x7f7f7f7f;

// ---- step 1+2: resized displacement, target cell, winner election
__global__ __launch_bounds__(kBlock) void iw_splat_kernel(
    const float* __restrict__ src_grid, const float* __restrict__ src_id, float* __restrict__ dxy,
    int* __restrict__ cell, int* __restrict__ winner, const int* __restrict__ rank, int Hs, int Ws,
    int H, int W) {
  const int64_t b = blockIdx.y;
  const int s = blockIdx.x * kBlock + threadIdx.x;
  const int HW = H * W;
  if (s >= HW) return;
  const int y = s / W, x = s - y * W;
  // PyTorch upsample_bilinear2d source index (align_corners=False)
  const float sh = (float)Hs / (float)H, sw = (float)Ws / (float)W;
  const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.0f);
  const float fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.0f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + ((y0 < Hs - 1) ? 1 : 0), x1 = x0 + ((x0 < Ws - 1) ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float* g = src_grid + b * Hs * Ws * 2;
  float d[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float v00 = g[(y0 * Ws + x0) * 2 + c] - src_id[(y0 * Ws + x0) * 2 + c];
    const float v01 = g[(y0 * Ws + x1) * 2 + c] - src_id[(y0 * Ws + x1) * 2 + c];
    const float v10 = g[(y1 * Ws + x0) * 2 + c] - src_id[(y1 * Ws + x0) * 2 + c];
    const float v11 = g[(y1 * Ws + x1) * 2 + c] - src_id[(y1 * Ws + x1) * 2 + c];
    d[c] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
  const float dx = d[0] * (float)W / 2.0f, dy = d[1] * (float)H / 2.0f;
  dxy[(b * 2 + 0) * HW + s] = dx;
  dxy[(b * 2 + 1) * HW + s] = dy;
  const float txf = rintf((float)x + dx), tyf = rintf((float)y + dy);  // round half to even
  int c = -1;
  if (txf >= 0.0f && tyf >= 0.0f && txf <= (float)(W - 1) && tyf <= (float)(H - 1)) c = (int)tyf * W + (int)txf;
  cell[b * HW + s] = c;
  // Election: the lowest priority (sample index, or position in the tie-break order) wins the cell.
  // An object map is a minification: runs of neighbouring samples land on one cell, and ~15 per
  // covered cell overall; the L2 arbitrates them one by one (the atomics are 3/4 of this kernel).
  // In sample-index order the first lane of a run of equal cells beats the rest of the run, so only
  // it asks.  (A look at the current winner before the atomic costs what the atomic costs: measured.)
  const int key = rank ? rank[s] : s;
  const int left = __shfl_up(c, 1, kWave);
  const bool run = rank == nullptr && (threadIdx.x & (kWave - 1)) != 0 && left == c;
  if (c >= 0 && !run) atomicMin(winner + b * HW + c, key);
}

// ---- step 2b (tie-break order given): the elected position -> the sample standing there
__global__ __launch_bounds__(kBlock) void iw_unrank_kernel(int* __restrict__ winner,
                                                           const int* __restrict__ order, int64_t n,
                                                           int HW) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int w = winner[i];
  if (w >= 0 && w < HW) winner[i] = order[w];
}

// ---- step 3a: padded field of the winners
__global__ __launch_bounds__(kBlock) void iw_gather_kernel(const float* __restrict__ dxy,
                                                           const int* __restrict__ winner,
                                                           float* __restrict__ field,
                                                           unsigned char* __restrict__ fill_iter,
                                                           int H, int W, int pad) {
  const int64_t b = blockIdx.y;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp, HW = H * W;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  const int yp = e / Wp, xp = e - yp * Wp;
  const int y = yp - pad, x = xp - pad;
  float vx = 0.0f, vy = 0.0f;
  unsigned char it = 255;
  if (y >= 0 && y < H && x >= 0 && x < W) {
    const int w = winner[b * HW + y * W + x];
    if (w != kNoWinner) {
      vx = -dxy[(b * 2 + 0) * HW + w];
      vy = -dxy[(b * 2 + 1) * HW + w];
      it = 0;
    }
  }
  field[(b * 2 + 0) * HWp + e] = vx;
  field[(b * 2 + 1) * HWp + e] = vy;
  fill_iter[b * HWp + e] = it;
}

// ---- step 3b: one Jacobi fill pass `iter` (1-based): cells filled so far have fill_iter < iter
__global__ __launch_bounds__(kBlock) void iw_fill_kernel(const float* __restrict__ fin,
                                                         float* __restrict__ fout,
                                                         unsigned char* __restrict__ fill_iter,
                                                         float* __restrict__ denom,
                                                         const float* __restrict__ kern, int Hp,
                                                         int Wp, int iter, int K) {
  const int64_t b = blockIdx.y;
  const int HWp = Hp * Wp;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  const int y = e / Wp, x = e - y * Wp;
  const unsigned char* fi = fill_iter + b * HWp;
  const float* fx = fin + (b * 2 + 0) * HWp;
  const float* fy = fin + (b * 2 + 1) * HWp;
  float vx = fx[e], vy = fy[e];
  if (fi[e] == 255) {
    // 4-neighbour ring of the set filled before this pass (neighbours outside the array: none)
    const bool up = y > 0 && fi[e - Wp] < iter, dn = y < Hp - 1 && fi[e + Wp] < iter;
    const bool lf = x > 0 && fi[e - 1] < iter, rt = x < Wp - 1 && fi[e + 1] < iter;
    if (up || dn || lf || rt) {
      // the K x K Gaussian-weighted mean of the filled neighbours (warp.py:140-146: conv2d with zero padding K / 2),
      // row by row -- for K == 3 the order iw_fused_kernel sums in
      const int r = K / 2;
      float sx = 0.0f, sy = 0.0f, sm = 0.0f;
      for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy >= 0 && yy < Hp && xx >= 0 && xx < Wp) {
            const int n = yy * Wp + xx;
            if (fi[n] < iter) {  // unfilled cells hold 0 and contribute nothing
              const float k = kern[(dy + r) * K + (dx + r)];
              sx = fmaf(k, fx[n], sx);
              sy = fmaf(k, fy[n], sy);
              sm += k;
            }
          }
        }
      vx = sx / sm;
      vy = sy / sm;
      denom[b * HWp + e] = sm;
    }
  }
  fout[(b * 2 + 0) * HWp + e] = vx;
  fout[(b * 2 + 1) * HWp + e] = vy;
}

// marks the cells filled in pass `iter` (separate launch: the pass reads fill_iter of neighbours)
__global__ __launch_bounds__(kBlock) void iw_mark_kernel(unsigned char* __restrict__ fill_iter,
                                                         const float* __restrict__ denom, int Hp,
                                                         int Wp, int iter) {
  const int64_t b = blockIdx.y;
  const int HWp = Hp * Wp;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  unsigned char* fi = fill_iter + b * HWp;
  if (fi[e] != 255) return;
  const int y = e / Wp, x = e - y * Wp;
  const bool up = y > 0 && fi[e - Wp] < iter, dn = y < Hp - 1 && fi[e + Wp] < iter;
  const bool lf = x > 0 && fi[e - 1] < iter, rt = x < Wp - 1 && fi[e + 1] < iter;
  // a neighbour marked in THIS launch carries `iter`, which is not < iter: no race on the decision
  if (up || dn || lf || rt) fi[e] = (unsigned char)iter;
}

// ---- step 4: one erosion pass on the mask (1 = filled)
__global__ __launch_bounds__(kBlock) void iw_erode_kernel(const unsigned char* __restrict__ min_,
                                                          unsigned char* __restrict__ mout, int Hp,
                                                          int Wp) {
  const int64_t b = blockIdx.y;
  const int HWp = Hp * Wp;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  const int y = e / Wp, x = e - y * Wp;
  const unsigned char* m = min_ + b * HWp;
  unsigned char v = m[e];
  if (v) {
    const bool hole = (y > 0 && !m[e - Wp]) || (y < Hp - 1 && !m[e + Wp]) || (x > 0 && !m[e - 1]) ||
                      (x < Wp - 1 && !m[e + 1]);
    if (hole) v = 0;
  }
  mout[b * HWp + e] = v;
}

__global__ __launch_bounds__(kBlock) void iw_mask_init_kernel(const unsigned char* __restrict__ fill_iter,
                                                              unsigned char* __restrict__ mask,
                                                              int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < n) mask[e] = fill_iter[e] != 255;
}

// ---- step 5
__global__ __launch_bounds__(kBlock) void iw_finalize_kernel(const float* __restrict__ field,
                                                             const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ tgt_id,
                                                             float* __restrict__ out, int H, int W,
                                                             int pad) {
  const int64_t b = blockIdx.y;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp, HW = H * W;
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= HW) return;
  const int y = s / W, x = s - y * W;
  const int e = (y + pad) * Wp + (x + pad);
  const bool m = mask[b * HWp + e] != 0;
  const float vx = m ? field[(b * 2 + 0) * HWp + e] : 2.0f * (float)W;
  const float vy = m ? field[(b * 2 + 1) * HWp + e] : 2.0f * (float)H;
  out[(b * HW + s) * 2 + 0] = tgt_id[s * 2 + 0] + vx * 2.0f / (float)W;
  out[(b * HW + s) * 2 + 1] = tgt_id[s * 2 + 1] + vy * 2.0f / (float)H;
}

// ---- steps 3-5 in one launch: a workgroup owns a 32 x 32 tile of the padded raster and redoes the
// fill / erosion passes on the tile plus a halo of 2 * niter cells in LDS (a cell's value depends on
// cells at most niter away, its eroded mask on fill states niter further).  Same arithmetic, in the
// same order, as the per-pass kernels above (which remain for niter too large for the LDS); instead of
// 3 * niter + 3 streaming passes over the padded raster, the winners are read once and the
// outputs written once.  A pass writes IN PLACE: the cells it fills are not read by anyone during
// the pass (readers only touch cells filled earlier), their fill state changes after a barrier.
constexpr int kFusedTH = 32, kFusedTW = 32;  // tile of the padded raster owned by a workgroup (32 x 64: slower)
constexpr size_t kFusedMaxLds = 65536;
constexpr unsigned char kOut = 254;  // outside the raster / outside the region: never filled, never a hole

__global__ __launch_bounds__(kBlock) void iw_fused_kernel(
    const float* __restrict__ dxy, const int* __restrict__ winner, const float* __restrict__ kern,
    const float* __restrict__ tgt_id, float* __restrict__ out, unsigned char* __restrict__ fill_iter,
    float* __restrict__ denom, unsigned char* __restrict__ mask, int H, int W, int niter, int erode,
    int tiles_x, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pad = niter + 1, halo = 2 * niter;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp, HW = H * W;
  const int RW = kFusedTW + 2 * halo + 2, RH = kFusedTH + 2 * halo + 2;  // region incl. a one-cell kOut border
  const int cells = RH * RW;
  float* fx = reinterpret_cast<float*>(smem);
  float* fy = fx + cells;
  unsigned char* fi = reinterpret_cast<unsigned char*>(fy + cells);
  unsigned char* m0 = fi + cells;
  unsigned char* m1 = m0 + cells;
  const int64_t b = blockIdx.x / tiles;
  const int tile = blockIdx.x % tiles;
  const int ty0 = (tile / tiles_x) * kFusedTH, tx0 = (tile % tiles_x) * kFusedTW;  // padded coords
  const int oy = ty0 - halo - 1, ox = tx0 - halo - 1;                                  // of region cell (0, 0)
  float k9[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k9[i] = kern[i];

  // ---- load: winners' (negated) displacement, fill state 0 / 255, kOut outside
  // Eight cells per thread and trip, in three sweeps: all the winners, then all the displacements, then the
  // LDS stores -- unconditional loads at clamped addresses, issued back to back (as one loop every cell was two
  // dependent memory round trips behind the previous cell's stores).
  int any_filled = 0;
  constexpr int kLB = 8;
  for (int c0 = threadIdx.x; c0 < cells; c0 += kBlock * kLB) {
    int wv[kLB];
    unsigned char stv[kLB];
    bool img[kLB];
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      const int c = min(c0 + u * kBlock, cells - 1);
      const int i = c / RW, j = c - i * RW;
      const int yp = oy + i, xp = ox + j;
      const bool inside = i > 0 && i < RH - 1 && j > 0 && j < RW - 1 && yp >= 0 && yp < Hp && xp >= 0 && xp < Wp;
      const int y = yp - pad, x = xp - pad;
      img[u] = inside && y >= 0 && y < H && x >= 0 && x < W;
      stv[u] = inside ? 255 : kOut;
      wv[u] = winner[b * HW + (img[u] ? y * W + x : 0)];
    }
    float vxv[kLB], vyv[kLB];
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      const bool has = img[u] && wv[u] != kNoWinner;
      const int w = has ? wv[u] : 0;
      vxv[u] = dxy[(b * 2 + 0) * HW + w];
      vyv[u] = dxy[(b * 2 + 1) * HW + w];
      if (has) stv[u] = 0;
    }
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      const int c = c0 + u * kBlock;
      if (c < cells) {
        const bool has = stv[u] == 0;
        fx[c] = has ? -vxv[u] : 0.0f;
        fy[c] = has ? -vyv[u] : 0.0f;
        fi[c] = stv[u];
        any_filled |= has;
      }
    }
  }
  // a region without a single winner stays empty through every pass (objects cover a few percent of
  // their canvas: most tiles): skip straight to the outputs
  const int n_pass = __syncthreads_or(any_filled) ? niter : 0;
  // ---- Jacobi fill passes.  The cells a pass fills (unfilled, with a 4-neighbour filled earlier) are few and
  // scattered -- a few hundred of the region's ~2900 -- so they are first COLLECTED into a list in LDS and the
  // 3 x 3 sums then run over the list with every lane busy.  Walking all cells with a branch around the sum,
  // a wave executed the sum's 27 LDS reads in every trip in which ANY of its lanes had such a cell: the LDS
  // pipe carried ~10x the reads needed and a full-coverage tile (the background's) took 60 us (timing
  // ablation: 96 of the kernel's 160 us were this loop).  Order inside a pass is free: a pass reads only cells
  // filled before it and writes only cells it fills; states change after the barrier.
  // (the two mask buffers, 2 * cells bytes, are not in use yet; fewer than cells - 1 cells can be on a ring)
  unsigned short* ring_list = reinterpret_cast<unsigned short*>(m0 + ((size_t)(m0 - smem) & 1));
  __shared__ int ring_count;
  // The collection itself on BIT ROWS where a region row has at most 64 cells (see the erosion below): one word per row
  // for the cells filled before the pass, one for the unfilled ones, and the ring of a pass is
  // `unfilled & (filled << 1 | filled >> 1 | filled above | filled below)` -- three word reads per row instead of five
  // byte reads per cell; a quarter row per thread then appends its set bits to the list.  The rows live behind the
  // list in the two mask buffers: a ring holds at most 4/5 of the region's interior (every ring cell needs a filled
  // neighbour, a filled cell has four), which leaves room for them from niter = 3 up (checked; else the byte scan).
  const bool bits = RW <= kWave;  // (uniform)
  unsigned long long* frow = reinterpret_cast<unsigned long long*>(m0 + ((2 * cells - 24 * RH) & ~7) - ((size_t)(m0 - smem) & 7));
  unsigned long long* urow = frow + RH;  // frow: filled before this pass; urow: unfilled; rrow: this pass's ring
  unsigned long long* rrow = urow + RH;
  const int ring_cap = (int)((reinterpret_cast<unsigned char*>(frow) - reinterpret_cast<unsigned char*>(ring_list)) / 2);
  const bool bit_rings = bits && 4 * RH <= kBlock && 5 * ring_cap >= 4 * (RW - 2) * (RH - 2) &&
                         reinterpret_cast<unsigned char*>(rrow + RH) <= m0 + 2 * cells;  // (uniform)
  if (bit_rings && n_pass > 0) {
    const int lane = threadIdx.x & (kWave - 1);
    for (int i = threadIdx.x >> 6; i < RH; i += kBlock / kWave) {
      const unsigned char f = lane < RW ? fi[i * RW + lane] : kOut;
      const unsigned long long fr = __ballot(f == 0), ur = __ballot(f == 255);
      if (lane == 0) {
        frow[i] = fr;
        urow[i] = ur;
      }
    }
  }
  for (int it = 1; it <= n_pass; ++it) {
    if (threadIdx.x == 0) ring_count = 0;
    __syncthreads();
    if (bit_rings) {
      const int i = threadIdx.x >> 2, q = threadIdx.x & 3;
      if (i > 0 && i < RH - 1) {  // (the border rows are kOut: never unfilled)
        const unsigned long long f = frow[i];
        const unsigned long long ring = urow[i] & ((f << 1) | (f >> 1) | frow[i - 1] | frow[i + 1]);
        if (q == 0) rrow[i] = ring;
        unsigned seg = (unsigned)(ring >> (16 * q)) & 0xffffu;
        if (seg) {
          int at = atomicAdd(&ring_count, __popc(seg));
          while (seg) {
            const int j = __ffs((int)seg) - 1;
            seg &= seg - 1;
            if (at < ring_cap) ring_list[at] = (unsigned short)(i * RW + 16 * q + j);  // (the bound holds: see above)
            ++at;
          }
        }
      } else if (q == 0 && i < RH) {
        rrow[i] = 0ull;
      }
    } else
    for (int c0 = threadIdx.x; c0 < cells; c0 += kBlock * kLB) {
      unsigned char s0[kLB], sn[kLB][4];
#pragma unroll
      for (int u = 0; u < kLB; ++u) s0[u] = fi[min(c0 + u * kBlock, cells - 1)];
#pragma unroll
      for (int u = 0; u < kLB; ++u) {
        const int c = min(c0 + u * kBlock, cells - 1);
        sn[u][0] = fi[max(c - RW, 0)];
        sn[u][1] = fi[min(c + RW, cells - 1)];
        sn[u][2] = fi[max(c - 1, 0)];
        sn[u][3] = fi[min(c + 1, cells - 1)];
      }
#pragma unroll
      for (int u = 0; u < kLB; ++u) {
        // (an unfilled cell, state 255, is never on the region's one-cell border: its neighbours exist)
        const bool on = c0 + u * kBlock < cells && s0[u] == 255 &&
                        (sn[u][0] < it || sn[u][1] < it || sn[u][2] < it || sn[u][3] < it);
        if (on) ring_list[atomicAdd(&ring_count, 1)] = (unsigned short)(c0 + u * kBlock);
      }
    }
    __syncthreads();
    const int nring = bit_rings ? min(ring_count, ring_cap) : ring_count;
    for (int e = threadIdx.x; e < nring; e += kBlock) {
      const int c = ring_list[e];
      unsigned char nf[9];
      float nx[9], ny[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int n = c + (k / 3 - 1) * RW + (k % 3 - 1);
        nf[k] = fi[n];
        nx[k] = fx[n];
        ny[k] = fy[n];
      }
      float sx = 0.0f, sy = 0.0f, sm = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const bool on = nf[k] < it;  // unfilled cells hold 0 and contribute nothing
        sx = on ? fmaf(k9[k], nx[k], sx) : sx;
        sy = on ? fmaf(k9[k], ny[k], sy) : sy;
        sm = on ? sm + k9[k] : sm;
      }
      fx[c] = sx / sm;
      fy[c] = sy / sm;
      const int i = c / RW, j = c - i * RW;
      const int yp = oy + i, xp = ox + j;
      if (yp >= ty0 && yp < ty0 + kFusedTH && xp >= tx0 && xp < tx0 + kFusedTW)
        denom[b * HWp + yp * Wp + xp] = sm;  // this workgroup's own cells (inside the raster: state 255)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nring; e += kBlock) fi[ring_list[e]] = (unsigned char)it;
    if (bit_rings && (int)threadIdx.x < RH) {  // the cells of this pass count as filled from the next one on
      const unsigned long long ring = rrow[threadIdx.x];
      frow[threadIdx.x] |= ring;
      urow[threadIdx.x] &= ~ring;
    }
  }
  __syncthreads();
  // ---- mask (1 = filled), erosion passes; kOut counts as filled (it is not part of the raster)
  // ROWS AS BITS where a region row has at most 64 cells (niter <= 7): row i of the mask is one 64-bit word (bit j =
  // cell (i, j)), a second word marks its kOut cells, and an erosion pass is `m & (kout | (up & down & m << 1 & m >> 1))`
  // on RH words -- the byte form below walks all ~2900 cells of the region per pass with six byte reads each
  // (timing ablation at the KITTI recipe: 0.55 of the inversion's 2.5 ms per pipeline step were these passes).
  // The same cells erode in the same passes: the same mask.  (The region's border cells are kOut: no row or column
  // outside the words is ever needed.)
  unsigned long long* rowm = reinterpret_cast<unsigned long long*>(m0 + ((8 - ((size_t)(m0 - smem) & 7)) & 7));
  unsigned long long* rowk = rowm + 2 * RH;  // [2][RH] mask rows in turn, [RH] kOut rows: 24 RH bytes of the 2 cells
  unsigned char* mi = m0;
  unsigned char* mo = m1;
  int cur = 0;
  if (bits) {
    const int lane = threadIdx.x & (kWave - 1);
    for (int i = threadIdx.x >> 6; i < RH; i += kBlock / kWave) {
      const unsigned char f = lane < RW ? fi[i * RW + lane] : (unsigned char)255;
      const unsigned long long mrow = __ballot(lane < RW && f != 255), krow = __ballot(lane < RW && f == kOut);
      if (lane == 0) {
        rowm[i] = mrow;
        rowk[i] = krow;
      }
    }
    if (erode) {
      for (int it = 0; it < n_pass; ++it) {
        __syncthreads();
        const int i = threadIdx.x;
        if (i < RH) {
          unsigned long long m = rowm[cur * RH + i];
          if (i > 0 && i < RH - 1)
            m &= rowk[i] | (rowm[cur * RH + i - 1] & rowm[cur * RH + i + 1] & (m << 1) & (m >> 1));
          rowm[(cur ^ 1) * RH + i] = m;
        }
        cur ^= 1;
      }
    }
  } else {
    for (int c = threadIdx.x; c < cells; c += kBlock) m0[c] = fi[c] != 255;
    if (erode) {
      for (int it = 0; it < n_pass; ++it) {
        __syncthreads();
        for (int c = threadIdx.x; c < cells; c += kBlock) {
          unsigned char v = mi[c];
          if (v && fi[c] != kOut && !(mi[c - RW] && mi[c + RW] && mi[c - 1] && mi[c + 1])) v = 0;
          mo[c] = v;
        }
        unsigned char* t = mi;
        mi = mo;
        mo = t;
      }
    }
  }
  __syncthreads();
  // ---- the tile's own cells: fill state, final mask, and (inside the image) the inverted grid
  for (int e = threadIdx.x; e < kFusedTH * kFusedTW; e += kBlock) {
    const int yp = ty0 + e / kFusedTW, xp = tx0 + e % kFusedTW;
    if (yp >= Hp || xp >= Wp) continue;
    const int c = (yp - oy) * RW + (xp - ox);
    fill_iter[b * HWp + yp * Wp + xp] = fi[c];
    const bool m = bits ? ((rowm[cur * RH + (yp - oy)] >> (xp - ox)) & 1ull) != 0 : mi[c] != 0;
    mask[b * HWp + yp * Wp + xp] = m ? 1 : 0;
    const int y = yp - pad, x = xp - pad;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const int s = y * W + x;
      const float vx = m ? fx[c] : 2.0f * (float)W;
      const float vy = m ? fy[c] : 2.0f * (float)H;
      out[(b * HW + s) * 2 + 0] = tgt_id[s * 2 + 0] + vx * 2.0f / (float)W;
      out[(b * HW + s) * 2 + 1] = tgt_id[s * 2 + 1] + vy * 2.0f / (float)H;
    }
  }
}

// ---- backward
__global__ __launch_bounds__(kBlock) void iw_bwd_init_kernel(const float* __restrict__ gout,
                                                             const unsigned char* __restrict__ mask,
                                                             float* __restrict__ gfield, int H, int W,
                                                             int pad) {
  const int64_t b = blockIdx.y;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp, HW = H * W;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  const int yp = e / Wp, xp = e - yp * Wp;
  const int y = yp - pad, x = xp - pad;
  float gx = 0.0f, gy = 0.0f;
  if (y >= 0 && y < H && x >= 0 && x < W && mask[b * HWp + e]) {
    gx = gout[(b * HW + y * W + x) * 2 + 0] * 2.0f / (float)W;
    gy = gout[(b * HW + y * W + x) * 2 + 1] * 2.0f / (float)H;
  }
  gfield[(b * 2 + 0) * HWp + e] = gx;
  gfield[(b * 2 + 1) * HWp + e] = gy;
}

// reverse of fill pass `iter`: cells filled earlier collect k * g / denom from the cells of this pass
__global__ __launch_bounds__(kBlock) void iw_bwd_fill_kernel(float* __restrict__ gfield,
                                                             const unsigned char* __restrict__ fill_iter,
                                                             const float* __restrict__ denom,
                                                             const float* __restrict__ kern, int Hp,
                                                             int Wp, int iter, int K) {
  const int64_t b = blockIdx.y;
  const int HWp = Hp * Wp;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= HWp) return;
  const unsigned char* fi = fill_iter + b * HWp;
  if (fi[e] >= iter) return;  // only earlier-filled cells fed pass `iter`
  const int y = e / Wp, x = e - y * Wp;
  float* gx = gfield + (b * 2 + 0) * HWp;
  float* gy = gfield + (b * 2 + 1) * HWp;
  const int r = K / 2;
  float ax = 0.0f, ay = 0.0f;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < Hp && xx >= 0 && xx < Wp) {
        const int n = yy * Wp + xx;
        if (fi[n] == iter) {
          // this cell sits at offset (-dy, -dx) in n's stencil
          const float k = kern[(r - dy) * K + (r - dx)] / denom[b * HWp + n];
          ax = fmaf(k, gx[n], ax);
          ay = fmaf(k, gy[n], ay);
        }
      }
    }
  gx[e] += ax;  // cells of pass `iter` are only read here, cells before it only written: in place
  gy[e] += ay;
}

// init + every reverse fill pass in ONE launch (round 3; the per-pass kernels above stay for iteration
// counts whose halo does not fit the LDS and as the reference the tests compare with): a workgroup owns a
// 32 x 32 tile of the padded raster and redoes the passes on the tile plus a halo of niter cells in LDS --
// the gradient a cell ends with depends on cells at most niter away (pass `it` moves gradient from the
// cells filled in pass `it` to their 3 x 3 neighbours filled earlier; passes run niter .. 1).  Same
// arithmetic in the same order as iw_bwd_init_kernel + iw_bwd_fill_kernel: bit-identical gfield.  At the
// LVD recipe the backward of the two grid inversions was 2 x (1 + 5) launches of 13-19 us each.
constexpr int kBwdTH = 32, kBwdTW = 32;

__global__ __launch_bounds__(kBlock) void iw_bwd_fused_kernel(
    const float* __restrict__ gout, const unsigned char* __restrict__ mask,
    const unsigned char* __restrict__ fill_iter, const float* __restrict__ denom,
    const float* __restrict__ kern, float* __restrict__ gfield, int H, int W, int niter, int tiles_x, int tiles,
    int sink_lists) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pad = niter + 1, halo = niter;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, HWp = Hp * Wp, HW = H * W;
  const int RW = kBwdTW + 2 * halo + 2, RH = kBwdTH + 2 * halo + 2;  // region incl. a one-cell border of state 255
  const int cells = RH * RW;
  float* gx = reinterpret_cast<float*>(smem);
  float* gy = gx + cells;
  float* rd = gy + cells;  // 1 / denom of the cell's own fill pass (0 for winners and unfilled cells)
  unsigned char* fi = reinterpret_cast<unsigned char*>(rd + cells);
  const int64_t b = blockIdx.x / tiles;
  const int tile = blockIdx.x % tiles;
  const int ty0 = (tile / tiles_x) * kBwdTH, tx0 = (tile % tiles_x) * kBwdTW;  // padded coords
  const int oy = ty0 - halo - 1, ox = tx0 - halo - 1;                              // of region cell (0, 0)
  float k9[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k9[i] = kern[i];
  // ---- load (= iw_bwd_init_kernel on the region): gradient of the masked image cells, fill states
  // (eight cells per thread and trip, every load unconditional at a clamped address and issued before the
  // first LDS store: see iw_fused_kernel)
  int any = 0;
  constexpr int kLB = 8;
  for (int c0 = threadIdx.x; c0 < cells; c0 += kBlock * kLB) {
    unsigned char stv[kLB], mk[kLB];
    float2 gv[kLB];
    float dv[kLB];
    bool inside[kLB], img[kLB];
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      const int c = min(c0 + u * kBlock, cells - 1);
      const int i = c / RW, j = c - i * RW;
      const int yp = oy + i, xp = ox + j;
      inside[u] = i > 0 && i < RH - 1 && j > 0 && j < RW - 1 && yp >= 0 && yp < Hp && xp >= 0 && xp < Wp;
      const int e = inside[u] ? yp * Wp + xp : 0;
      const int y = yp - pad, x = xp - pad;
      img[u] = inside[u] && y >= 0 && y < H && x >= 0 && x < W;
      stv[u] = fill_iter[b * HWp + e];
      mk[u] = mask[b * HWp + e];
      dv[u] = denom[b * HWp + e];
      gv[u] = *reinterpret_cast<const float2*>(gout + (b * HW + (img[u] ? y * W + x : 0)) * 2);
    }
#pragma unroll
    for (int u = 0; u < kLB; ++u) {
      const int c = c0 + u * kBlock;
      if (c < cells) {
        // outside: state 255, never a source (fi == it) nor a sink (fi < it) of any pass
        const unsigned char st = inside[u] ? stv[u] : (unsigned char)255;
        const bool g = img[u] && mk[u];
        gx[c] = g ? gv[u].x * 2.0f / (float)W : 0.0f;
        gy[c] = g ? gv[u].y * 2.0f / (float)H : 0.0f;
        rd[c] = (st >= 1 && st <= niter) ? dv[u] : 1.0f;
        fi[c] = st;
        any |= (st >= 1 && st <= niter);
      }
    }
  }
  // a region without a single filled cell: nothing moves
  const int n_pass = __syncthreads_or(any) ? niter : 0;
  // what a cell filled before pass `it` collects from the cells of pass `it` around it
  auto collect = [&](int c, int it) {
    float ax = 0.0f, ay = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int n = c + dy * RW + dx;
        if (fi[n] == it) {
          // this cell sits at offset (-dy, -dx) in n's stencil; the Gaussian is symmetric
          const float k = k9[(1 - dy) * 3 + (1 - dx)] / rd[n];
          ax = fmaf(k, gx[n], ax);
          ay = fmaf(k, gy[n], ay);
        }
      }
    gx[c] += ax;  // cells of pass `it` are only read here, cells before it only written: in place
    gy[c] += ay;
  };
  if (sink_lists && n_pass > 0) {
    // The cells that collect anything in pass `it` -- filled before it, with a cell of pass `it` among their eight
    // neighbours -- are few: they come from BIT ROWS (one 64-bit word per region row and fill state, as the forward
    // kernel's rings) into a list, and the 3 x 3 sums run over the list with every lane busy.  Walking all cells, every
    // earlier-filled one (most of a background region) paid nine byte reads per pass to find nothing.  A cell left
    // off the list would have added 0.
    unsigned long long* srow = reinterpret_cast<unsigned long long*>(smem + (((size_t)cells * 13 + 7) & ~(size_t)7));
    unsigned short* list = reinterpret_cast<unsigned short*>(srow + (niter + 1) * RH);
    __shared__ int list_count;
    const int lane = threadIdx.x & (kWave - 1);
    for (int i = threadIdx.x >> 6; i < RH; i += kBlock / kWave) {
      const unsigned char f = lane < RW ? fi[i * RW + lane] : (unsigned char)255;
      for (int k = 0; k <= niter; ++k) {
        const unsigned long long row = __ballot(f == k);
        if (lane == 0) srow[k * RH + i] = row;
      }
    }
    for (int it = n_pass; it >= 1; --it) {
      if (threadIdx.x == 0) list_count = 0;
      __syncthreads();
      const int i = threadIdx.x >> 2, q = threadIdx.x & 3;
      if (i > 0 && i < RH - 1) {
        unsigned long long before = 0ull, around = 0ull;
        for (int k = 0; k < it; ++k) before |= srow[k * RH + i];
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
          const unsigned long long sr = srow[it * RH + i + d];
          around |= sr | (sr << 1) | (sr >> 1);
        }
        unsigned seg = (unsigned)((before & around) >> (16 * q)) & 0xffffu;
        if (seg) {
          int at = atomicAdd(&list_count, __popc(seg));
          while (seg) {
            const int j = __ffs((int)seg) - 1;
            seg &= seg - 1;
            list[at++] = (unsigned short)(i * RW + 16 * q + j);  // (at most `cells` entries: the buffer's size)
          }
        }
      }
      __syncthreads();
      const int nl = list_count;
      for (int e = threadIdx.x; e < nl; e += kBlock) collect(list[e], it);
      __syncthreads();
    }
  } else {
    for (int it = n_pass; it >= 1; --it) {
      for (int c = threadIdx.x; c < cells; c += kBlock) {
        if (fi[c] >= it) continue;  // only earlier-filled cells fed pass `it` (border cells: 255)
        collect(c, it);
      }
      __syncthreads();
    }
  }
  // ---- the tile's own cells
  for (int e = threadIdx.x; e < kBwdTH * kBwdTW; e += kBlock) {
    const int yp = ty0 + e / kBwdTW, xp = tx0 + e % kBwdTW;
    if (yp >= Hp || xp >= Wp) continue;
    const int c = (yp - oy) * RW + (xp - ox);
    gfield[(b * 2 + 0) * HWp + yp * Wp + xp] = gx[c];
    gfield[(b * 2 + 1) * HWp + yp * Wp + xp] = gy[c];
  }
}

// d loss / d (dx, dy) of the winners, then the adjoint of the bilinear resize -- as a GATHER: one
// thread per source texel and component pair sums the samples whose bilinear footprint contains the
// texel (those with y0 == j or y1 == j, likewise in x: a (2 H / Hs) x (2 W / Ws) window), each with
// the weight the forward gave it.  No atomics, no zero-fill, bitwise reproducible; the scatter it
// replaces put 8 float atomics per winner on addresses shared by the 4-8 neighbouring samples
// (68 us per call at the recipe).
__global__ __launch_bounds__(kBlock) void iw_bwd_gather_kernel(
    const float* __restrict__ gfield, const int* __restrict__ cell, const int* __restrict__ winner,
    float* __restrict__ gsrc, int Hs, int Ws, int H, int W, int pad) {
  const int64_t b = blockIdx.y;
  const int HW = H * W, Wp = W + 2 * pad, HWp = (H + 2 * pad) * Wp;
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= Hs * Ws) return;
  const int j = t / Ws, i = t - j * Ws;
  const float sh = (float)Hs / (float)H, sw = (float)Ws / (float)W;
  // samples whose source row interval [y0, y1] can contain j: fy in (j - 1, j + 1), one sample of
  // slack on either side for the rounding of the float expressions (the weights below are exact
  // zeros / skips for a sample that does not touch the texel)
  const int ya = max((int)floorf(((float)j - 0.5f) / sh - 0.5f) - 1, 0);
  const int yb = min((int)ceilf(((float)j + 1.5f) / sh - 0.5f) + 1, H - 1);
  const int xa = max((int)floorf(((float)i - 0.5f) / sw - 0.5f) - 1, 0);
  const int xb = min((int)ceilf(((float)i + 1.5f) / sw - 0.5f) + 1, W - 1);
  float ax = 0.0f, ay = 0.0f;
  // rows / columns of the window that do not touch the texel are skipped before any load (walking
  // the whole window for winners first and weighting only those measured 2x slower: the loads of
  // `cell`, 16 bytes apart across the lanes, are what this kernel pays for)
  for (int y = ya; y <= yb; ++y) {
    const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, y1 = y0 + ((y0 < Hs - 1) ? 1 : 0);
    if (y0 != j && y1 != j) continue;
    const float ly = fy - (float)y0;
    // weight of row j in this sample (y0 == y1 at the last row: both terms go to it, as in the forward)
    const float wy = (y0 == j ? 1.0f - ly : 0.0f) + (y1 == j ? ly : 0.0f);
    for (int x = xa; x <= xb; ++x) {
      const float fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.0f);
      const int x0 = (int)fx, x1 = x0 + ((x0 < Ws - 1) ? 1 : 0);
      if (x0 != i && x1 != i) continue;
      const int s = y * W + x;
      const int c = cell[b * HW + s];
      if (c < 0 || winner[b * HW + c] != s) continue;
      const float lx = fx - (float)x0;
      const float wx = (x0 == i ? 1.0f - lx : 0.0f) + (x1 == i ? lx : 0.0f);
      const int cy = c / W, cx = c - cy * W;
      const int e = (cy + pad) * Wp + (cx + pad);
      // field = -d(px);  d(px) = d(normalised) * size / 2
      const float gdx = -gfield[(b * 2 + 0) * HWp + e] * (float)W / 2.0f;
      const float gdy = -gfield[(b * 2 + 1) * HWp + e] * (float)H / 2.0f;
      ax = fmaf(wy * wx, gdx, ax);
      ay = fmaf(wy * wx, gdy, ay);
    }
  }
  float* g = gsrc + (b * Hs * Ws + t) * 2;
  g[0] = ax;
  g[1] = ay;
}

static int check_iw(const char* fn, int64_t B, int Hs, int Ws, int H, int W, int niter, int ksize) {
  if (B < 0 || Hs < 1 || Ws < 1 || H < 1 || W < 1 || niter < 0 || niter > 200 || B > 65535 ||
      (int64_t)(H + 2 * niter + 2) * (W + 2 * niter + 2) > 2147483647 / 4) {
    set_error("%s: bad shape B=%lld src=%dx%d tgt=%dx%d niter=%d (B <= 65535, niter <= 200)", fn,
              (long long)B, Hs, Ws, H, W, niter);
    return WALDO_EINVAL;
  }
  if (ksize < 1 || ksize > 15 || ksize % 2 == 0) {
    set_error("%s: kernel size %d (odd, 1 ... 15: the reference's conv2d with padding K / 2 keeps the raster for odd K only)",
              fn, ksize);
    return WALDO_EINVAL;
  }
  return WALDO_OK;
}

}  // namespace waldo

using namespace waldo;

namespace waldo {
static int inverse_warp_fwd_impl(const char* fn, const float* src_grid, const float* src_id,
                                 const float* tgt_id, const float* gauss, float* out, float* dxy,
                                 int* cell, int* winner, float* field_a, float* field_b,
                                 unsigned char* fill_iter, float* denom, unsigned char* mask_a,
                                 unsigned char* mask_b, const int* rank, const int* order, int64_t B,
                                 int Hs, int Ws, int H, int W, int niter, int erode, int ksize,
                                 waldo_stream_t stream);
}

extern "C" int waldo_inverse_warp_fwd(const float* src_grid, const float* src_id,
                                      const float* tgt_id, const float* gauss, float* out,
                                      float* dxy, int* cell, int* winner, float* field_a,
                                      float* field_b, unsigned char* fill_iter, float* denom,
                                      unsigned char* mask_a, unsigned char* mask_b, int64_t B,
                                      int Hs, int Ws, int H, int W, int niter, int erode, int ksize,
                                      waldo_stream_t stream) {
  return inverse_warp_fwd_impl("waldo_inverse_warp_fwd", src_grid, src_id, tgt_id, gauss, out,
                               dxy, cell, winner, field_a, field_b, fill_iter, denom, mask_a, mask_b,
                               nullptr, nullptr, B, Hs, Ws, H, W, niter, erode, ksize, stream);
}

extern "C" int waldo_inverse_warp_order_fwd(const float* src_grid, const float* src_id,
                                            const float* tgt_id, const float* gauss,
                                            const int* rank, const int* order, float* out,
                                            float* dxy, int* cell, int* winner, float* field_a,
                                            float* field_b, unsigned char* fill_iter, float* denom,
                                            unsigned char* mask_a, unsigned char* mask_b, int64_t B,
                                            int Hs, int Ws, int H, int W, int niter, int erode, int ksize,
                                            waldo_stream_t stream) {
  if (B > 0 && (!rank || !order)) {
    set_error("waldo_inverse_warp_order_fwd: null pointer");
    return WALDO_EINVAL;
  }
  return inverse_warp_fwd_impl("waldo_inverse_warp_order_fwd", src_grid, src_id, tgt_id, gauss,
                               out, dxy, cell, winner, field_a, field_b, fill_iter, denom, mask_a,
                               mask_b, rank, order, B, Hs, Ws, H, W, niter, erode, ksize, stream);
}

int waldo::inverse_warp_fwd_impl(const char* fn, const float* src_grid, const float* src_id,
                                        const float* tgt_id, const float* gauss, float* out,
                                        float* dxy, int* cell, int* winner, float* field_a,
                                        float* field_b, unsigned char* fill_iter, float* denom,
                                        unsigned char* mask_a, unsigned char* mask_b,
                                        const int* rank, const int* order, int64_t B, int Hs, int Ws,
                                        int H, int W, int niter, int erode, int ksize, waldo_stream_t stream) {
  int rc = check_iw(fn, B, Hs, Ws, H, W, niter, ksize);
  if (rc) return rc;
  if (B == 0) return WALDO_OK;
  if (!src_grid || !src_id || !tgt_id || !gauss || !out || !dxy || !cell || !winner ||
      !field_a || !field_b || !fill_iter || !denom || !mask_a || !mask_b) {
    set_error("%s: null pointer", fn);
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int pad = niter + 1;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int HW = H * W, HWp = Hp * Wp;
  dim3 gs((HW + kBlock - 1) / kBlock, (unsigned)B), gp((HWp + kBlock - 1) / kBlock, (unsigned)B);
  // every cell starts without a winner (a fill KERNEL: see fill_words in waldo_common.hip.h)
  fill_words(winner, (unsigned)kNoWinner, sizeof(int) * (size_t)B * HW, st);
  hipLaunchKernelGGL(iw_splat_kernel, gs, dim3(kBlock), 0, st, src_grid, src_id, dxy, cell, winner,
                     rank, Hs, Ws, H, W);
  if (order) {
    const int64_t nw = (int64_t)B * HW;
    hipLaunchKernelGGL(iw_unrank_kernel, dim3((unsigned)((nw + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       st, winner, order, nw, HW);
  }
  const bool passes = debug_option(WALDO_DEBUG_IW_PASSES);
  const int tiles_x = (Wp + kFusedTW - 1) / kFusedTW, tiles_y = (Hp + kFusedTH - 1) / kFusedTH;
  const size_t lds = (size_t)(kFusedTH + 4 * niter + 2) * (kFusedTW + 4 * niter + 2) * (2 * sizeof(float) + 3);
  // (the one-launch kernel is written for the 3 x 3 kernel every script uses; other sizes take the passes one by one)
  if (!passes && ksize == 3 && lds <= kFusedMaxLds && (int64_t)B * tiles_x * tiles_y <= 2147483647) {
    hipLaunchKernelGGL(iw_fused_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), dim3(kBlock), lds, st, dxy,
                       winner, gauss, tgt_id, out, fill_iter, denom, mask_a, H, W, niter, erode, tiles_x,
                       tiles_x * tiles_y);
    return launch_status(fn);
  }
  hipLaunchKernelGGL(iw_gather_kernel, gp, dim3(kBlock), 0, st, dxy, winner, field_a, fill_iter, H,
                     W, pad);
  float* fin = field_a;
  float* fout = field_b;
  for (int it = 1; it <= niter; ++it) {
    hipLaunchKernelGGL(iw_fill_kernel, gp, dim3(kBlock), 0, st, fin, fout, fill_iter, denom,
                       gauss, Hp, Wp, it, ksize);
    hipLaunchKernelGGL(iw_mark_kernel, gp, dim3(kBlock), 0, st, fill_iter, denom, Hp, Wp, it);
    float* t = fin;
    fin = fout;
    fout = t;
  }
  const int64_t n = (int64_t)B * HWp;
  hipLaunchKernelGGL(iw_mask_init_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, st, fill_iter, mask_a, n);
  unsigned char* mi = mask_a;
  unsigned char* mo = mask_b;
  if (erode) {
    for (int it = 0; it < niter; ++it) {
      hipLaunchKernelGGL(iw_erode_kernel, gp, dim3(kBlock), 0, st, mi, mo, Hp, Wp);
      unsigned char* t = mi;
      mi = mo;
      mo = t;
    }
  }
  // the final mask is left in mask_a for the backward
  if (mi != mask_a)
    copy_bytes(mask_a, mi, (size_t)n, st);
  hipLaunchKernelGGL(iw_finalize_kernel, gs, dim3(kBlock), 0, st, fin, mask_a, tgt_id, out, H, W,
                     pad);
  return launch_status(fn);
}

extern "C" int waldo_inverse_warp_bwd(const float* grad_out, const float* gauss,
                                      const int* cell, const int* winner,
                                      const unsigned char* fill_iter, const float* denom,
                                      const unsigned char* mask, float* gfield,
                                      float* grad_src_grid, int64_t B, int Hs, int Ws, int H, int W,
                                      int niter, int ksize, waldo_stream_t stream) {
  int rc = check_iw("waldo_inverse_warp_bwd", B, Hs, Ws, H, W, niter, ksize);
  if (rc) return rc;
  if (B == 0) return WALDO_OK;
  if (!grad_out || !gauss || !cell || !winner || !fill_iter || !denom || !mask || !gfield ||
      !grad_src_grid) {
    set_error("waldo_inverse_warp_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const int pad = niter + 1;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int HW = H * W, HWp = Hp * Wp;
  dim3 gs((HW + kBlock - 1) / kBlock, (unsigned)B), gp((HWp + kBlock - 1) / kBlock, (unsigned)B);
  const int bw_rw = kBwdTW + 2 * niter + 2, bw_rh = kBwdTH + 2 * niter + 2;
  size_t bw_lds = (size_t)bw_rw * bw_rh * 13;
  // (the rows of bits and the list of iw_bwd_fused_kernel's passes, where a region row fits a 64-bit word, a quarter
  // row a thread, and the LDS the lot)
  const size_t bw_lists = (((size_t)bw_rw * bw_rh * 13 + 7) & ~(size_t)7) + (size_t)(niter + 1) * bw_rh * 8 +
                          (size_t)bw_rw * bw_rh * 2;
  const int sink_lists = bw_rw <= 64 && 4 * bw_rh <= kBlock && bw_lists <= kFusedMaxLds;
  if (sink_lists) bw_lds = bw_lists;
  const int btx = (Wp + kBwdTW - 1) / kBwdTW, bty = (Hp + kBwdTH - 1) / kBwdTH;
  if (!debug_option(WALDO_DEBUG_IW_PASSES) && ksize == 3 && bw_lds <= kFusedMaxLds && B * btx * bty <= 2147483647ll) {
    hipLaunchKernelGGL(iw_bwd_fused_kernel, dim3((unsigned)(B * btx * bty)), dim3(kBlock), bw_lds, st, grad_out,
                       mask, fill_iter, denom, gauss, gfield, H, W, niter, btx, btx * bty, sink_lists);
  } else {
    hipLaunchKernelGGL(iw_bwd_init_kernel, gp, dim3(kBlock), 0, st, grad_out, mask, gfield, H, W, pad);
    for (int it = niter; it >= 1; --it)
      hipLaunchKernelGGL(iw_bwd_fill_kernel, gp, dim3(kBlock), 0, st, gfield, fill_iter, denom,
                         gauss, Hp, Wp, it, ksize);
  }
  hipLaunchKernelGGL(iw_bwd_gather_kernel, dim3((Hs * Ws + kBlock - 1) / kBlock, (unsigned)B), dim3(kBlock), 0,
                     st, gfield, cell, winner, grad_src_grid, Hs, Ws, H, W, pad);
  return launch_status("waldo_inverse_warp_bwd");
}
