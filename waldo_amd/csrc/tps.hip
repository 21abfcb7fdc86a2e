// A2: thin-plate-spline grid synthesis (replaces TPSWarp.forward, models/modules/warp.py:49-55).
//   mapping = K^-1 @ [src_pts; 0]   (tiny; one thread per output element)
//   grid    = basis @ mapping       (output-bound: 8 B written per pixel per map)
// The basis is stored transposed (K3, HW): lane i of a wavefront reads pixel p0+i of basis
// function k, one coalesced 256-B request per k.
#include "waldo_common.hip.h"

namespace waldo {

__global__ __launch_bounds__(kBlock) void tps_mapping_fwd_kernel(
    const float* __restrict__ inv, const float* __restrict__ pts, float* __restrict__ mapping,
    int64_t B, int N) {
  const int K3 = N + 3;
  const int64_t total = B * K3 * 2;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(e & 1);
    const int r = (int)((e >> 1) % K3);
    const int64_t b = (e >> 1) / K3;
    const float* row = inv + (int64_t)r * K3;
    const float* x = pts + b * N * 2 + c;
    // the fma chain keeps its order (bit-identical); unrolled so that eight steps' loads are in
    // flight instead of one dependent round trip per control point (N = 128 for the background)
    float acc = 0.0f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) acc = fmaf(row[n], x[2 * n], acc);
    mapping[e] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void tps_mapping_bwd_kernel(
    const float* __restrict__ inv, const float* __restrict__ gmap, float* __restrict__ gpts,
    int64_t B, int N) {
  const int K3 = N + 3;
  const int64_t total = B * N * 2;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(e & 1);
    const int n = (int)((e >> 1) % N);
    const int64_t b = (e >> 1) / N;
    const float* g = gmap + b * K3 * 2 + c;
    float acc = 0.0f;
#pragma unroll 8
    for (int r = 0; r < K3; ++r) acc = fmaf(inv[(int64_t)r * K3 + n], g[2 * r], acc);
    gpts[e] = acc;
  }
}

constexpr int kGridNB = 8;  // maps evaluated per thread per pass over the basis

__global__ __launch_bounds__(kBlock) void tps_grid_fwd_kernel(const float* __restrict__ basis_t,
                                                              const float* __restrict__ mapping,
                                                              float* __restrict__ grid, int64_t B,
                                                              int64_t HW, int K3) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = p < HW;
  const int64_t pc = live ? p : HW - 1;
  const int64_t b0 = (int64_t)blockIdx.y * kGridNB;
  const int nb = (int)min((int64_t)kGridNB, B - b0);
  float ax[kGridNB], ay[kGridNB];
#pragma unroll
  for (int i = 0; i < kGridNB; ++i) ax[i] = ay[i] = 0.0f;
#pragma unroll 8  // eight basis loads in flight (K3 = 131 for the background: the loop was one round trip per k)
  for (int k = 0; k < K3; ++k) {
    const float bv = basis_t[(int64_t)k * HW + pc];
#pragma unroll
    for (int i = 0; i < kGridNB; ++i) {
      if (i < nb) {  // wave-uniform
        const float* m = mapping + ((b0 + i) * K3 + k) * 2;
        ax[i] = fmaf(bv, m[0], ax[i]);
        ay[i] = fmaf(bv, m[1], ay[i]);
      }
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < kGridNB; ++i) {
      if (i < nb) {
        float2* o = reinterpret_cast<float2*>(grid + ((b0 + i) * HW + p) * 2);
        *o = make_float2(ax[i], ay[i]);
      }
    }
  }
}

// The same with the workgroup's kGridNB mappings staged in LDS ([k][map][xy]: the 16 floats of a k are four
// broadcast ds_read_b128).  Read through the scalar cache -- mapping + ((b0 + i) * K3 + k) * 2 with K3 in a
// register -- they were 2 K3 kGridNB scalar loads per wavefront with their address arithmetic: at the LVD
// recipe's background (K3 = 131) 1578 scalar against 989 vector instructions per wave, 43 us for a 17 MB
// stream.  Same fma chain per output: same bits.
constexpr int kGridLdsMaxK = 768;  // 48 KB of LDS

__global__ __launch_bounds__(kBlock) void tps_grid_fwd_lds_kernel(const float* __restrict__ basis_t,
                                                                  const float* __restrict__ mapping,
                                                                  float* __restrict__ grid, int64_t B,
                                                                  int64_t HW, int K3) {
  typedef float f32x4_g __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smap[];  // [K3][kGridNB][2]
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = p < HW;
  const int64_t pc = live ? p : HW - 1;
  const int64_t b0 = (int64_t)blockIdx.y * kGridNB;
  const int nb = (int)min((int64_t)kGridNB, B - b0);
  for (int e = threadIdx.x; e < K3 * kGridNB * 2; e += kBlock) {
    const int k = e / (kGridNB * 2), r = e - k * (kGridNB * 2), i = r >> 1;
    smap[e] = i < nb ? mapping[((b0 + i) * K3 + k) * 2 + (r & 1)] : 0.0f;
  }
  __syncthreads();
  f32x4_g acc[kGridNB / 2];  // {x, y} of two maps each
#pragma unroll
  for (int i = 0; i < kGridNB / 2; ++i) acc[i] = (f32x4_g){0.0f, 0.0f, 0.0f, 0.0f};
  const float* bp = basis_t + pc;
#pragma unroll 8  // eight basis loads in flight
  for (int k = 0; k < K3; ++k) {
    const float bv = bp[(int64_t)k * HW];
    const f32x4_g* m = reinterpret_cast<const f32x4_g*>(smap + k * (kGridNB * 2));
#pragma unroll
    for (int i = 0; i < kGridNB / 2; ++i) {
      const f32x4_g mv = m[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = fmaf(bv, mv[c], acc[i][c]);
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < kGridNB; ++i) {
      if (i < nb) {
        float2* o = reinterpret_cast<float2*>(grid + ((b0 + i) * HW + p) * 2);
        *o = make_float2(acc[i >> 1][2 * (i & 1)], acc[i >> 1][2 * (i & 1) + 1]);
      }
    }
  }
}

constexpr int kGradNB = 4;    // maps per workgroup
constexpr int kGradPPT = 4;   // pixels per thread and chunk (summed in registers before the wave reduce)
constexpr int kGradKB = 4;    // basis functions per wave reduction
constexpr int kGradMaxK = 136;  // K3 values whose partial sums fit the workgroup's LDS table

// grad_mapping[b][k][c] = sum_p basis_t[k][p] * grad_grid[b][p][c]: a skinny contraction over the
// pixels.  A workgroup takes kGradNB maps and walks `chunks` chunks of 1024 pixels; per chunk and k
// every wave reduces its pixels (wave_transpose_reduce: lane bitrev(idx) ends up with entry idx of
// the 2 kGradNB partial sums) and adds the result to ITS OWN row of an LDS table; at the end the
// four rows are summed in a fixed order and leave the workgroup as ONE float atomic per (map, k,
// component) -- the first version issued one per wave and chunk on a few hundred addresses, the
// pattern measured 14x below the streaming atomic rate.  K3 > kGradMaxK: per-wave atomics as before.
__global__ __launch_bounds__(kBlock) void tps_grid_bwd_kernel(const float* __restrict__ basis_t,
                                                              const float* __restrict__ ggrid,
                                                              float* __restrict__ gmap, int64_t B,
                                                              int64_t HW, int K3, int chunks, int kper) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  // blockIdx.z: a range of kper basis functions (a multiple of kGradKB).  The background grids of the LVD recipe are 10
  // maps of 32768 pixels against K3 = 131: 96 workgroups, each a chain of 33 dependent load -> fma -> reduce rounds
  // (59 us); with the basis functions dealt over five workgroups the chains are 7 rounds long
  const int kbeg = (int)blockIdx.z * kper, kend = min(K3, kbeg + kper);
  const int64_t b0 = (int64_t)blockIdx.y * kGradNB;
  const int nb = (int)min((int64_t)kGradNB, B - b0);
  __shared__ float acc[4][kGradMaxK * kGradNB * 2];
  const bool table = K3 <= kGradMaxK;
  if (table)
    for (int e = kbeg * kGradNB * 2 + lane; e < kend * kGradNB * 2; e += kWave) acc[wave][e] = 0.0f;  // wave-private row
  for (int c = 0; c < chunks; ++c) {
    const int64_t pbase = (((int64_t)blockIdx.x * chunks + c) * kBlock + threadIdx.x) * kGradPPT;
    if ((((int64_t)blockIdx.x * chunks + c) * kBlock) * kGradPPT >= HW) break;  // block-uniform
    float g[kGradPPT][kGradNB][2];
#pragma unroll
    for (int q = 0; q < kGradPPT; ++q) {
      const int64_t p = pbase + q;
#pragma unroll
      for (int i = 0; i < kGradNB; ++i) {
        if (p < HW && i < nb) {
          const float2 v = *reinterpret_cast<const float2*>(ggrid + ((b0 + i) * HW + p) * 2);
          g[q][i][0] = v.x;
          g[q][i][1] = v.y;
        } else {
          g[q][i][0] = g[q][i][1] = 0.0f;
        }
      }
    }
    // kGradKB basis functions per wave reduction: one transpose-reduce of 32 partial sums instead of four of
    // 8 (half the shuffles, a quarter of the dependent reduction chains: at K3 = 131 those chains, one per k,
    // were what the kernel's 68 us were made of); their sixteen basis loads are in flight together
    // (unconditional: past the raster the gradient is 0).  Each partial sum is the same fma chain as before.
    for (int k0 = kbeg; k0 < kend; k0 += kGradKB) {
      float part[kGradKB * kGradNB * 2];
#pragma unroll
      for (int i = 0; i < kGradKB * kGradNB * 2; ++i) part[i] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < kGradKB; ++kk) {
        const int k = min(k0 + kk, K3 - 1);  // (a repeated k past the end is computed and dropped)
#pragma unroll
        for (int q = 0; q < kGradPPT; ++q) {
          const int64_t p = pbase + q;
          const float bv = basis_t[(int64_t)k * HW + (p < HW ? p : HW - 1)];
#pragma unroll
          for (int i = 0; i < kGradNB; ++i) {
            part[kk * kGradNB * 2 + 2 * i] = fmaf(bv, g[q][i][0], part[kk * kGradNB * 2 + 2 * i]);
            part[kk * kGradNB * 2 + 2 * i + 1] = fmaf(bv, g[q][i][1], part[kk * kGradNB * 2 + 2 * i + 1]);
          }
        }
      }
      const float red = wave_transpose_reduce<kGradKB * kGradNB * 2>(part, lane);
      const int e = bitrev6(lane);
      const int k = k0 + e / (kGradNB * 2), idx = e % (kGradNB * 2);
      if (e < kGradKB * kGradNB * 2 && k < kend && idx < nb * 2) {
        if (table)
          acc[wave][k * kGradNB * 2 + idx] += red;  // one lane per entry: plain read-modify-write
        else
          atomicAdd(gmap + ((b0 + (idx >> 1)) * K3 + k) * 2 + (idx & 1), red);
      }
    }
  }
  if (table) {
    __syncthreads();
    for (int e = kbeg * kGradNB * 2 + threadIdx.x; e < kend * kGradNB * 2; e += kBlock) {
      const int k = e / (kGradNB * 2), idx = e % (kGradNB * 2);
      if (idx < nb * 2)
        atomicAdd(gmap + ((b0 + (idx >> 1)) * K3 + k) * 2 + (idx & 1),
                  (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]));
    }
  }
}

static int grid_for(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  return (int)max((int64_t)1, min(blocks, (int64_t)256 * 16));
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_tps_mapping_fwd(const float* inverse_kernel, const float* src_pts,
                                     float* mapping, int64_t B, int N, waldo_stream_t stream) {
  if (B < 0 || N < 1) {
    set_error("waldo_tps_mapping_fwd: bad shape B=%lld N=%d", (long long)B, N);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!inverse_kernel || !src_pts || !mapping) {
    set_error("waldo_tps_mapping_fwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(tps_mapping_fwd_kernel, dim3(grid_for(B * (N + 3) * 2)), dim3(kBlock), 0,
                     (hipStream_t)stream, inverse_kernel, src_pts, mapping, B, N);
  return launch_status("waldo_tps_mapping_fwd");
}

extern "C" int waldo_tps_mapping_bwd(const float* inverse_kernel, const float* grad_mapping,
                                     float* grad_src_pts, int64_t B, int N,
                                     waldo_stream_t stream) {
  if (B < 0 || N < 1) {
    set_error("waldo_tps_mapping_bwd: bad shape B=%lld N=%d", (long long)B, N);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!inverse_kernel || !grad_mapping || !grad_src_pts) {
    set_error("waldo_tps_mapping_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipLaunchKernelGGL(tps_mapping_bwd_kernel, dim3(grid_for(B * N * 2)), dim3(kBlock), 0,
                     (hipStream_t)stream, inverse_kernel, grad_mapping, grad_src_pts, B, N);
  return launch_status("waldo_tps_mapping_bwd");
}

extern "C" int waldo_tps_grid_fwd(const float* basis_t, const float* mapping, float* grid,
                                  int64_t B, int64_t HW, int K3, waldo_stream_t stream) {
  if (B < 0 || HW < 1 || K3 < 3 || (B + kGridNB - 1) / kGridNB > 65535) {
    set_error("waldo_tps_grid_fwd: bad shape B=%lld HW=%lld K3=%d", (long long)B, (long long)HW,
              K3);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!basis_t || !mapping || !grid) {
    set_error("waldo_tps_grid_fwd: null pointer");
    return WALDO_EINVAL;
  }
  dim3 g((unsigned)((HW + kBlock - 1) / kBlock), (unsigned)((B + kGridNB - 1) / kGridNB));
  if (K3 <= kGridLdsMaxK)
    hipLaunchKernelGGL(tps_grid_fwd_lds_kernel, g, dim3(kBlock), sizeof(float) * K3 * kGridNB * 2,
                       (hipStream_t)stream, basis_t, mapping, grid, B, HW, K3);
  else
    hipLaunchKernelGGL(tps_grid_fwd_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, basis_t,
                       mapping, grid, B, HW, K3);
  return launch_status("waldo_tps_grid_fwd");
}

extern "C" int waldo_tps_grid_bwd(const float* basis_t, const float* grad_grid,
                                  float* grad_mapping, int64_t B, int64_t HW, int K3,
                                  waldo_stream_t stream) {
  if (B < 0 || HW < 1 || K3 < 3 || (B + kGradNB - 1) / kGradNB > 65535) {
    set_error("waldo_tps_grid_bwd: bad shape B=%lld HW=%lld K3=%d", (long long)B, (long long)HW,
              K3);
    return WALDO_EINVAL;
  }
  if (B == 0) return WALDO_OK;
  if (!basis_t || !grad_grid || !grad_mapping) {
    set_error("waldo_tps_grid_bwd: null pointer");
    return WALDO_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  fill_words(grad_mapping, 0u, sizeof(float) * B * K3 * 2, st);
  const int64_t per_chunk = (int64_t)kBlock * kGradPPT;
  const int64_t nchunks = (HW + per_chunk - 1) / per_chunk, groups_b = (B + kGradNB - 1) / kGradNB;
  // several chunks per workgroup (fewer atomics per output) while keeping >= ~512 workgroups
  const int chunks = (int)min((int64_t)16, max((int64_t)1, nchunks * groups_b / 512));
  // few workgroups (few, large maps): the basis functions are dealt over blockIdx.z, in multiples of kGradKB
  const int64_t wgs = ((nchunks + chunks - 1) / chunks) * groups_b;
  const int rounds = (K3 + kGradKB - 1) / kGradKB;
  const int ksplit = (int)min((int64_t)rounds, max((int64_t)1, 512 / wgs));
  const int kper = ((rounds + ksplit - 1) / ksplit) * kGradKB;
  dim3 g((unsigned)((nchunks + chunks - 1) / chunks), (unsigned)groups_b, (unsigned)((K3 + kper - 1) / kper));
  hipLaunchKernelGGL(tps_grid_bwd_kernel, g, dim3(kBlock), 0, st, basis_t, grad_grid,
                     grad_mapping, B, HW, K3, chunks, kper);
  return launch_status("waldo_tps_grid_bwd");
}
