// fp32 MFMA implicit-GEMM convolution engine for gfx950 (CDNA4).
//
// One stride-1 window correlation over a logical NCDHW view; Conv2d / Conv3d /
// ConvTranspose3d forward + data-gradient + weight-gradient of the reference
// (resnet50.py:62-69,95-107,124; reconstruction_decoder.py:49-95;
// ray_traced_skip_connection.py:38) all map onto it through views and packed
// weights built by corenet_amd/model/conv_geometry.py.
//
// Design (MI355X-first, not a cuDNN/im2col translation):
//  * A workgroup (4 waves) owns a TDxTHxTW tile of output positions and an
//    N-tile of NSUB*16 output channels.  For each chunk of CC input channels it
//    stages the raw input PATCH (tile + window halo) and the weight chunk in
//    LDS once, then walks the window taps: every tap is an LDS *offset*, not a
//    new gather, so each HBM/L2 byte of the patch feeds kd*kh*kw*N MACs.
//  * v_mfma_f32_16x16x4_f32: rows = 16 output positions (an mh x mw sub-tile),
//    cols = 16 output channels, k = 4 input channels at one tap.  Exact fp32
//    (the reference is fp32 end to end), 157 TF/s peak.
//  * Register-staged software pipeline: the global loads of chunk c+1 are
//    issued into VGPRs before the MFMA loop of chunk c and written to LDS after
//    it, so HBM/L2 latency hides under the matrix pipe (cdna guide T14).
//  * LDS strides are padded to 16 (mod 32) banks so the two 32-lane halves of a
//    ds_read_b32 (k = 0,1 / 2,3) never collide.
//  * BatchRenorm-apply + ReLU of the producer are fused into the patch staging
//    (crnInTransform), so normalised activations are never written to HBM.
#include "conv_kernels.h"
#include <algorithm>
#include <mutex>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
using namespace crnk;

// --------------------------- pointwise (1x1x1) convolutions ---------------------------------
// Y[b][n][m] = bias[n] + sum_c W[c][n] * T(X[b][c][m]) for plain views with contiguous positions:
// a GEMM whose A operand is already K-major in memory (NCHW), so global loads are float4 rows
// and no index arithmetic is needed.  Block = 64 positions x 32 channels, K chunks of 32 staged
// through LDS with register prefetch; wave w owns positions [16w, 16w+16) x 32 channels.
// Used for ResNet 1x1 convs and their data gradients (resnet50.py:62-69,95-107) and the skip
// compress convs (ray_traced_skip_connection.py:38).
struct PwGeom {
  const float* x; float* y; const float* w; const float* bias;
  crnInTransform tr;
  int B, C, N, Npad, S;            // S = positions per sample
  int64_t xsB, ysB;                // batch strides; x channel stride = S
  int64_t ysC, ysP;                // y channel / position strides: (S, 1) plain, (1, row pitch) channel-last
  int bias_sB, mode;
  int splits, cps;                 // split-K: blockIdx.z = split*B + b, cps input channels per split (multiple
                                   // of 32); partial sums go to the dense scratch y (mode 0), split*B + b as batch
  // fused reduction (counters != nullptr): the last workgroup of a tile adds the splits up and writes yr
  float* yr; int64_t yr_sB, yr_sC, yr_sP; int yr_accumulate; int* counters;
  long long* stamps;               // tuning aid (CRN_PW_STAMPS=1): shader-clock stamps of workgroup 0
  int ablate;                      // tuning aid (CRN_PW_ABLATE bits): 1 no MFMAs, 2 no global loads, 4 no stores, 8 no LDS commit
};

// NS = 16-column blocks per workgroup (2: 64 x 32 tiles; 4: 64 x 64 tiles -- half the workgroups and half the
// re-reads of x for the wide layers)
constexpr int kPwTab = 2048;        // channels of one workgroup (its split's range) whose scale / shift are staged in LDS
template <int NS>
__global__ __launch_bounds__(256) void pointwise_fwd_kernel(PwGeom g) {
  constexpr int BM = 64, BN = NS * 16, KC = 64, SA = BM + 16, SB = BN + 16;   // KC 64: half the barriers of KC 32
  constexpr int NRA = KC / 16;
  __shared__ __attribute__((aligned(16))) float ldsA[KC * SA];
  __shared__ __attribute__((aligned(16))) float ldsB[KC * SB];
  crn_kernargs_now(g.x, g.y, g.w, g.bias, g.tr.scale, g.tr.shift, g.tr.pre_relu, g.tr.post_relu, g.B, g.C, g.N, g.Npad,
                   g.S, g.xsB, g.ysB, g.ysC, g.ysP, g.bias_sB, g.mode, g.splits, g.cps, g.counters);
  const long long t_entry = (long long)__builtin_amdgcn_s_memtime();
  const long long r_entry = (long long)__builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool stamp = g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
  int nmark = 0;
  auto mark = [&]() { if (stamp && nmark < 30) g.stamps[nmark] = (long long)__builtin_amdgcn_s_memtime() - t_entry; ++nmark; };
  mark();
  const int split = blockIdx.z / g.B, b = blockIdx.z - split * g.B;
  const int cbeg = split * g.cps, cend = min(g.C, cbeg + g.cps);
  const float* xb = g.x + (int64_t)b * g.xsB;
  // staging roles: A: KC/16 float4 per thread (rows ka, ka+16, ...), B: NRB float4 per thread
  const int ka = tid >> 4, ca = (tid & 15) * 4;       // A row / column
  constexpr int BQ = BN / 4, BROWS = 256 / BQ, NRB = KC / BROWS;   // B: float4 per row, rows per pass, passes (1 or 2)
  const int kb = tid / BQ, cb = (tid % BQ) * 4;       // B row / column
  const bool a_ok = (m0 + ca) < g.S;                  // S % 4 == 0 (checked on the host)
  const bool b_ok = (n0 + cb) < g.Npad;
  // Two chunks of loads are in flight (a chunk's MFMAs take 0.45 us, a load 1-2 us: with one chunk in flight every
  // iteration waited out most of a memory latency), hand-tracked like in conv_e2d.hip: asm buffer loads, s_waitcnt
  // vmcnt(N) with the N the issue order implies, out-of-range offsets (zeros, no traffic) past the end.  The
  // BatchRenorm scale / shift of the workgroup's channels are staged into LDS once, next to the first loads (read from
  // global inside the loop they cost another exposed latency per chunk).
  // (dynamic LDS, 2 * cps floats: static tables for 2048 channels took a third of the kernel's LDS and with it the
  // fourth and fifth resident workgroup -- and residency is what these latency chains are short of, see the host side)
  extern __shared__ float pw_tables[];
  const bool has_tr = g.tr.scale != nullptr;
  float* tsc = pw_tables; float* tsh = pw_tables + g.cps;
  constexpr unsigned kOOBo = 0x80000000u;
  constexpr int NL = NRA + NRB;                      // loads per thread and chunk
  const crn_rsrc xrs = crnk::make_rsrc(xb);
  const crn_rsrc wrs = crnk::make_rsrc(g.w);
  f32x4 ra[2][NRA], rb[2][NRB];
  auto issue = [&](int c0, f32x4 (&pa_)[NRA], f32x4 (&pb_)[NRB]) {
#pragma unroll
    for (int q = 0; q < NRA; ++q) {
      const int c_a = c0 + ka + q * 16;
      const unsigned off = (a_ok && c_a < cend && !(g.ablate & 2)) ? (unsigned)(c_a * g.S + m0 + ca) * 4u : kOOBo;
      crnk::crn_bload4(pa_[q], xrs, off);
    }
#pragma unroll
    for (int q = 0; q < NRB; ++q) {
      const int c_b = c0 + kb + q * BROWS;
      const unsigned off = (b_ok && c_b < cend && !(g.ablate & 2)) ? (unsigned)(c_b * g.Npad + n0 + cb) * 4u : kOOBo;
      crnk::crn_bload4(pb_[q], wrs, off);
    }
  };
  auto wait_slot = [&](f32x4 (&pa_)[NRA], f32x4 (&pb_)[NRB]) {   // the other slot's NL loads were issued later
#pragma unroll
    for (int q = 0; q < NRA; ++q) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pa_[q]) : "n"(NL));
#pragma unroll
    for (int q = 0; q < NRB; ++q) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pb_[q]) : "n"(NL));
  };
  auto xform = [&](f32x4 v, int c) -> f32x4 {
    if (has_tr && a_ok && c < cend) {
      const float sc = tsc[c - cbeg], sh = tsh[c - cbeg];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = v[i];
        if (g.tr.pre_relu) t = fmaxf(t, 0.f);
        t = t * sc + sh;
        if (g.tr.post_relu) t = fmaxf(t, 0.f);
        v[i] = t;
      }
    }
    return v;
  };
  f32x4 acc[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) acc[ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
  issue(cbeg, ra[0], rb[0]);
  issue(cbeg + KC, ra[1], rb[1]);
  mark();
  if (has_tr)
    for (int c = tid; c < cend - cbeg; c += 256) { tsc[c] = g.tr.scale[cbeg + c]; tsh[c] = g.tr.shift[cbeg + c]; }
  auto step = [&](int c0, f32x4 (&pa_)[NRA], f32x4 (&pb_)[NRB]) {
    wait_slot(pa_, pb_);
    mark();
    if (!(g.ablate & 8)) {
    __syncthreads();                                   // (first step: the tables; later: the previous chunk's MFMA reads)
#pragma unroll
    for (int q = 0; q < NRA; ++q) *reinterpret_cast<f32x4*>(ldsA + (ka + q * 16) * SA + ca) = xform(pa_[q], c0 + ka + q * 16);
#pragma unroll
    for (int q = 0; q < NRB; ++q) *reinterpret_cast<f32x4*>(ldsB + (kb + q * BROWS) * SB + cb) = pb_[q];
    __syncthreads();
    }
    mark();
    issue(c0 + 2 * KC, pa_, pb_);                      // this slot's registers are free: the chunk after the next
    if (g.ablate & 1) { mark(); return; }
    const float* pa = ldsA + kk * SA + wave * 16 + i16;
    const float* pb = ldsB + kk * SB + i16;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const float a = pa[ks * 4 * SA];
      float bv[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bv[ns] = pb[ks * 4 * SB + ns * 16];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[ns], acc[ns], 0, 0, 0);
    }
    mark();
  };
  for (int c0 = cbeg; c0 < cend; c0 += 2 * KC) {
    step(c0, ra[0], rb[0]);
    if (c0 + KC < cend) step(c0 + KC, ra[1], rb[1]);
  }
  // the loads issued past the end (zeros) still target the staging registers: keep them allocated until they landed
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
    for (int q = 0; q < NRA; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[sl][q]));
#pragma unroll
    for (int q = 0; q < NRB; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[sl][q]));
  }
  // D: rows kk*4..kk*4+3 = 4 consecutive positions, col i16 = channel -> one float4 per lane
  const int m = m0 + wave * 16 + kk * 4;
  if (m < g.S && !(g.ablate & 4)) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int n = n0 + ns * 16 + i16;
      if (n < g.N) {
        const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
        float* dst = g.y + (int64_t)(split * g.B + b) * g.ysB + (int64_t)n * g.ysC + (int64_t)m * g.ysP;
        f32x4 v = acc[ns] + bsv;
        if (g.ysP == 1) {
          if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {          // channel-last output: the 16 lanes of a row group write 16 consecutive channels
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[i * g.ysP] = g.mode == 1 ? dst[i * g.ysP] + v[i] : v[i];
        }
      }
    }
  }
  if (stamp) {
    g.stamps[31] = (long long)__builtin_amdgcn_s_memtime() - t_entry;
    g.stamps[30] = (long long)__builtin_amdgcn_s_memrealtime() - r_entry;      // 100 MHz: cycles / this = shader clock
  }
  if (g.counters) {       // fused split-K reduction: see conv_fwd_kernel (mode 4)
    __threadfence();
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
      const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * b);
      const int ticket = atomicAdd(g.counters + id, 1);
      const int last = ticket == g.splits - 1;
      if (last) g.counters[id] = 0;
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (m < g.S) {
      const int64_t slab = (int64_t)g.B * g.ysB;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int n = n0 + ns * 16 + i16;
        if (n < g.N) {
          const float* src = g.y + (int64_t)b * g.ysB + (int64_t)n * g.ysC + m;      // scratch: dense [n][pos]
          f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
          for (int sp = 0; sp < g.splits; ++sp) sum += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + sp * slab));
          float* dst = g.yr + (int64_t)b * g.yr_sB + (int64_t)n * g.yr_sC + (int64_t)m * g.yr_sP;
          if (g.yr_sP == 1) {
            if (g.yr_accumulate) sum += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = sum;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * g.yr_sP] = g.yr_accumulate ? dst[i * g.yr_sP] + sum[i] : sum[i];
          }
        }
      }
    }
  }
}

// --------------------------- pointwise convolutions, operands straight from HBM -------------------------------
// The kernel above stages K chunks of 64 channels through LDS, two barriers per chunk, and every chunk waits out most
// of a memory latency (~2 us on a busy chip): 11-19 us per launch for layers whose FLOPs and bytes are worth 3-5 us
// -- 80 launches, 1.1 ms of a 8 ms training step (profiles/r02_bench_kernel_stats.csv, its top line).  This one has no
// staging at all:
//   * NCHW activations are K-major for the A operand and the packed weights [Cin][Npad] for the B operand once the
//     16 rows / 16 columns of an MFMA tile are taken with stride 4: lane (i16, kk) loads ONE float4 = positions
//     p0 + 4*i16 .. +3 of channel c + kk, and element j of it is row i16 of M tile j (positions p0 + 4*i + j); the
//     same for the weights (columns n0 + 4*i + jn).  One A float4 and one B float4 per lane feed 16
//     v_mfma_f32_16x16x4_f32 (4 M tiles x 4 N tiles, K = 4 channels): a wave owns a 64 position x 64 column tile, all
//     loads are full 256-byte rows, and nothing goes through LDS on the way in.
//   * Work split: a workgroup = NW waves on one 64 x 64 tile, wave w takes the 16-channel groups w, w + NW, ... of
//     the workgroup's K slice (GPW groups per wave, all their loads issued at once: one memory latency per
//     workgroup), then the NW partial tiles are added up through LDS (one barrier) and stored as float4 rows.
//     Layers with few tiles split K over workgroups as well (partial sums to the split-K scratch, summed by the
//     BatchRenorm launch that follows -- crn_splitk_defer -- or by the reduction launch).
//   * The BatchRenorm-apply + ReLU of the producer is applied to the A registers (scale / shift of the lane's channel).
// fp32 MFMA like the kernel above: same products, different summation order (2e-5 of the output range in the tests).
struct Pw2Geom {
  const float* x; const float* w; const float* bias; float* y;
  crnInTransform tr;
  int B, C, N, Npad, S, tilesS;
  int64_t xsB, ysB, ysC;
  int bias_sB, mode;           // mode 1: y += result
  int ksl, ksplits;            // channels per workgroup slice (multiple of 16), slices; slices > 1: y is the scratch,
};                             // batch index split * B + b, no bias except slice 0

template <int NW, int GPW>
__global__ __launch_bounds__(NW * 64) void pw2_kernel(Pw2Geom g) {
  crn_kernargs_now(g.x, g.w, g.bias, g.y, g.tr.scale, g.tr.shift, g.tr.pre_relu, g.tr.post_relu, g.B, g.C, g.N, g.Npad,
                   g.S, g.tilesS, g.xsB, g.ysB, g.ysC, g.bias_sB, g.mode, g.ksl, g.ksplits);
  extern __shared__ __attribute__((aligned(16))) f32x4 red[];        // [NW][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  const int b = blockIdx.x / g.tilesS, p0 = (blockIdx.x - b * g.tilesS) * 64;
  const int n0 = blockIdx.y * 64;
  const int split = blockIdx.z;
  const int cbeg = split * g.ksl, cend = min(g.C, cbeg + g.ksl);
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + (int64_t)b * g.xsB), 0, g.C * g.S * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0, g.C * g.Npad * 4, 0x00020000);
  const bool p_ok = p0 + 4 * i16 < g.S, n_ok = n0 + 4 * i16 < g.Npad;
  const bool has_tr = g.tr.scale != nullptr;
  f32x4 av[GPW][4], bv[GPW][4];
  float sc[GPW][4], sh[GPW][4];
  constexpr unsigned kOut = 0x80000000u;                             // past the buffer: reads 0
#pragma unroll
  for (int t = 0; t < GPW; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = cbeg + 16 * (wave + NW * t) + 4 * q + kk;
      const bool ok = c < cend;
      av[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          xrs, (ok && p_ok) ? (unsigned)(c * g.S + p0 + 4 * i16) * 4u : kOut, 0, 0));
      bv[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          wrs, (ok && n_ok) ? (unsigned)(c * g.Npad + n0 + 4 * i16) * 4u : kOut, 0, 0));
      sc[t][q] = (has_tr && ok) ? g.tr.scale[c] : 1.f;
      sh[t][q] = (has_tr && ok) ? g.tr.shift[c] : 0.f;
    }
  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) acc[j][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < GPW; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a = av[t][q];
      if (has_tr) {
        const int c = cbeg + 16 * (wave + NW * t) + 4 * q + kk;
        if (c < cend && p_ok) {                                      // (channels / positions past the end stay 0)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = a[j];
            if (g.tr.pre_relu) v = fmaxf(v, 0.f);
            v = v * sc[t][q] + sh[t][q];
            if (g.tr.post_relu) v = fmaxf(v, 0.f);
            a[j] = v;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[j][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv[t][q][jn], acc[j][jn], 0, 0, 0);
    }
  // D row kk*4 + r of M tile j = position p0 + 4*(kk*4 + r) + j, col i16 of N tile jn = column n0 + 4*i16 + jn: the four
  // M tiles of one (jn, r) are four consecutive positions -> one float4 per (jn, r) and lane, in output order
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      red[(wave * 16 + jn * 4 + r) * 64 + lane] = (f32x4){acc[0][jn][r], acc[1][jn][r], acc[2][jn][r], acc[3][jn][r]};
  __syncthreads();
  constexpr int EPW = 16 / NW;                                       // (jn, r) pairs summed and stored by each wave
#pragma unroll
  for (int e = 0; e < EPW; ++e) {
    const int pr = wave * EPW + e, jn = pr >> 2, r = pr & 3;
    f32x4 v = red[pr * 64 + lane];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) v += red[(w2 * 16 + pr) * 64 + lane];
    const int n = n0 + 4 * i16 + jn, p = p0 + 16 * kk + 4 * r;
    if (n < g.N && p < g.S) {
      if (g.bias && split == 0) v += g.bias[(int64_t)b * g.bias_sB + n];
      float* dst = g.y + (int64_t)(split * g.B + b) * g.ysB + (int64_t)n * g.ysC + p;
      if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
}

template <int NW, int GPW>
int launch_pw2(const Pw2Geom& g, dim3 grid, hipStream_t st) {
  auto k = pw2_kernel<NW, GPW>;
  const size_t lds = (size_t)NW * 16 * 64 * 16;
  if (lds > 65536) CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

// --------------------------- dense layers on 1^3 grids ------------------------------------------------------
// decoder stage_1 (reconstruction_decoder.py:52-54): ConvTranspose3d(k=4) from a 1^3 grid = a dense layer from
// 67 channels onto 16384 logical channels, i.e. ONE position per sample.  The tile engine needs 16 positions per
// MFMA row block and spent 67 / 76 / 47 us (forward / data gradient / weight gradient) on 1.1 M multiply-adds per
// sample; these three loops are bandwidth bound on the 4.4 MB of weights instead.
constexpr int kDenseB = 8;      // samples per launch
// many output columns, few input channels: a workgroup owns 64 columns, its four waves split the input channels (one
// thread per column and 256 columns per workgroup left 192 CUs idle and walked the 67 rows in three batches of loads:
// 25 us for 4.4 MB of weights); the weight rows of a thread are in flight before the inputs are staged
__global__ __launch_bounds__(256) void dense_cols_kernel(const float* x, int64_t xsB, int64_t xsC, crnInTransform tr,
                                                         const float* w, int Npad, int N, int C, int B,
                                                         const float* bias, int bias_sB, float* y, int64_t ysB,
                                                         int64_t ysC, int accumulate) {
  extern __shared__ float xs[];                        // [B][C] transformed inputs
  __shared__ float red[3][kDenseB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int cq = (C + 3) >> 2, cbeg = wave * cq, cend = min(C, cbeg + cq);
  const bool col = n < N;
  float wv[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) wv[u] = (col && cbeg + u < cend) ? w[(int64_t)(cbeg + u) * Npad + n] : 0.f;
  for (int i = threadIdx.x; i < B * C; i += blockDim.x) {
    const int b = i / C, c = i - b * C;
    float v = x[b * xsB + c * xsC];
    if (tr.scale) {
      if (tr.pre_relu) v = fmaxf(v, 0.f);
      v = v * tr.scale[c] + tr.shift[c];
      if (tr.post_relu) v = fmaxf(v, 0.f);
    }
    xs[i] = v;
  }
  __syncthreads();
  float acc[kDenseB];
#pragma unroll
  for (int b = 0; b < kDenseB; ++b) acc[b] = 0.f;
  for (int c = cbeg; c < cend; c += 32) {
    if (c > cbeg) {
#pragma unroll
      for (int u = 0; u < 32; ++u) wv[u] = (col && c + u < cend) ? w[(int64_t)(c + u) * Npad + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u)
#pragma unroll
      for (int b = 0; b < kDenseB; ++b)
        if (b < B && c + u < cend) acc[b] += xs[b * C + c + u] * wv[u];
  }
  if (wave) {
#pragma unroll
    for (int b = 0; b < kDenseB; ++b) red[wave - 1][b][lane] = acc[b];
  }
  __syncthreads();
  if (wave || !col) return;
#pragma unroll
  for (int b = 0; b < kDenseB; ++b)
    if (b < B) {
      float* d = y + b * ysB + n * ysC;
      const float v = ((acc[b] + red[0][b][lane]) + (red[1][b][lane] + red[2][b][lane])) +
                      (bias ? bias[(int64_t)b * bias_sB + n] : 0.f);
      *d = accumulate ? *d + v : v;
    }
}
// few output columns, many input channels: a workgroup owns (sample b, 8 columns), its 256 threads stride over the
// channels (two float4 of weights per channel), and the 256 partial sums of every column are added in the workgroup
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* x, int64_t xsB, int64_t xsC, const float* w,
                                                         int Npad, int N, int C, const float* bias, int bias_sB,
                                                         float* y, int64_t ysB, int64_t ysC, int accumulate,
                                                         int cps) {
  // blockIdx.z = slice of cps input channels (gridDim.z > 1: y was zeroed / holds the accumulation base and every
  // slice adds its part atomically -- 36 workgroups walking 32768 rows each took 47 us)
  __shared__ float red[4][8];
  const int b = blockIdx.y, n0 = blockIdx.x * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float* xb = x + b * xsB;
  const int cbeg = blockIdx.z * cps, cend = min(C, cbeg + cps);
#pragma unroll 4
  for (int c = cbeg + threadIdx.x; c < cend; c += 256) {
    const float xv = xb[c * xsC];
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + (int64_t)c * Npad + n0);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + (int64_t)c * Npad + n0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] += xv * w0[j]; acc[4 + j] += xv * w1[j]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = crn_wave_sum(acc[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8 && n0 + (int)threadIdx.x < N) {
    const int n = n0 + threadIdx.x;
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (bias && blockIdx.z == 0) v += bias[(int64_t)b * bias_sB + n];
    float* d = y + b * ysB + n * ysC;
    if (gridDim.z > 1) atomicAdd(d, v);
    else *d = accumulate ? *d + v : v;
  }
}
// weight gradient: dw[c][n] += sum_b T(x)[b][c] * dy[b][n]
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* x, int64_t xsB, int64_t xsC, crnInTransform tr,
                                                          const float* dy, int64_t dsB, int64_t dsC, float* dw,
                                                          int Npad, int N, int C, int B) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (n >= N) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    float v = x[b * xsB + c * xsC];
    if (tr.scale) {
      if (tr.pre_relu) v = fmaxf(v, 0.f);
      v = v * tr.scale[c] + tr.shift[c];
      if (tr.post_relu) v = fmaxf(v, 0.f);
    }
    acc += v * dy[b * dsB + n * dsC];
  }
  dw[(int64_t)c * Npad + n] += acc;
}

bool one_position(const crnView& v) { return v.D == 1 && v.H == 1 && v.W == 1 && v.chan_off == nullptr; }

// 16-byte staging: unit W stride and every offset a multiple of 4 floats
bool vec_view(const crnView& v) {
  return v.chan_off == nullptr && v.sW == 1 && (v.W & 3) == 0 && (v.sH & 3) == 0 && (v.sD & 3) == 0 &&
         (v.sC & 3) == 0 && (v.sB & 3) == 0 && (((uintptr_t)v.base) & 15) == 0;
}
int vec_lead(int pw) { return ((-pw) % 4 + 4) % 4; }   // (w0 - pw - lead) % 4 == 0 for 4-aligned tile origins

bool plain_view(const crnView& v) {
  return v.chan_off == nullptr && v.sW == 1 && v.sH == v.W && (v.D == 1 || v.sD == v.H * v.W) &&
         v.sC == (int64_t)v.D * v.H * v.W && (((uintptr_t)v.base) & 15) == 0 && (v.sB & 3) == 0;
}

// y of a pointwise conv: positions flatten to one index with a single stride (plain: 1; channel-last
// [B][pos][C]: the pixel pitch), channels have their own stride
bool flat_out_view(const crnView& v) {
  if (plain_view(v)) return true;
  return v.chan_off == nullptr && v.sH == (int64_t)v.W * v.sW && (v.D == 1 || v.sD == (int64_t)v.H * v.W * v.sW) &&
         v.sC >= 1 && v.sW >= 1;
}

__global__ void zero_view_kernel(crnView v) {
  const int64_t per_b = (int64_t)v.C * v.D * v.H * v.W;
  const int64_t total = per_b * v.B;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int w = r % v.W; r /= v.W;
    const int h = r % v.H; r /= v.H;
    const int d = r % v.D; r /= v.D;
    const int c = r % v.C; r /= v.C;
    const int64_t co = v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
    v.base[r * v.sB + co + (int64_t)d * v.sD + (int64_t)h * v.sH + (int64_t)w * v.sW] = 0.f;
  }
}

// y[b,c,pos] (view) = (accumulate ? y : 0) + sum_s scratch[s][b][c][pos] (dense)
__global__ void splitk_reduce_kernel(crnView v, const float* scratch, int splits, int accumulate) {
  const int64_t S = (int64_t)v.D * v.H * v.W, per_b = (int64_t)v.C * S;
  const int64_t total = per_b * v.B;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int w = r % v.W; r /= v.W;
    const int h = r % v.H; r /= v.H;
    const int d = r % v.D; r /= v.D;
    const int c = r % v.C; r /= v.C;
    float sum = 0.f;
    for (int s = 0; s < splits; ++s) sum += scratch[(int64_t)s * total + e];
    const int64_t co = v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
    float* dst = v.base + r * v.sB + co + (int64_t)d * v.sD + (int64_t)h * v.sH + (int64_t)w * v.sW;
    *dst = accumulate ? *dst + sum : sum;
  }
}

// the same reduction for plain outputs (every sample a contiguous [C][D][H][W] block): float4 elements, no index
// arithmetic, and the loads of four splits in flight at a time (the generic kernel walks its splits one load latency
// after the other and spends five integer divisions per element; 60 of these launches sit on the step's critical path)
__global__ __launch_bounds__(256) void splitk_reduce_plain_kernel(float* y, int64_t ysB, const float* scratch,
                                                                  int64_t per_b4, int64_t total4, int splits,
                                                                  int accumulate) {
  crn_kernargs_now(y, ysB, scratch, per_b4, total4, splits, accumulate);
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(scratch);
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
      const f32x4 a = sc4[(int64_t)s * total4 + e], b = sc4[(int64_t)(s + 1) * total4 + e];
      const f32x4 c = sc4[(int64_t)(s + 2) * total4 + e], d = sc4[(int64_t)(s + 3) * total4 + e];
      sum += a; sum += b; sum += c; sum += d;               // same order as the generic kernel: split 0 first
    }
    for (; s < splits; ++s) sum += sc4[(int64_t)s * total4 + e];
    const int64_t b = e / per_b4, r = e - b * per_b4;
    f32x4* dst = reinterpret_cast<f32x4*>(y + b * ysB) + r;
    *dst = accumulate ? *dst + sum : sum;
  }
}

// per-device scratch for split-K partial sums (one process drives one GPU; calls that use it on different
// streams of the same device must not overlap)
// One scratch per (device, stream): convolutions that split their reduction on DIFFERENT streams (the engine runs the
// ray-traced skip path beside the encoder / decoder chain) must not share partial-sum storage.  Grown on demand;
// hipFree synchronises the device, so a buffer is never released under a kernel that still uses it.
constexpr int kSkSlots = 64;
// pinned: the slot belongs to a stream whose launches are CAPTURED into HIP graphs (crn_splitk_reserve): a graph holds the
// buffer's address, so a pinned slot that has to grow keeps its outgrown buffers (`retired`) until crn_splitk_release.
struct SkSlot { int dev; hipStream_t st; float* buf; size_t cap; bool used; bool pinned; std::vector<float*>* retired; };
SkSlot g_sk_slots[kSkSlots] = {};
std::mutex g_sk_mu;                      // the table is shared by every host thread that launches (autograd worker, main)
float* splitk_scratch(size_t floats, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  constexpr int kSlots = kSkSlots;
  SkSlot* const slots = g_sk_slots;
  typedef SkSlot Slot;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  Slot* sl = nullptr;
  for (int i = 0; i < kSlots && !sl; ++i)
    if (slots[i].used && slots[i].dev == dev && slots[i].st == st) sl = &slots[i];
  for (int i = 0; i < kSlots && !sl; ++i)
    if (!slots[i].used) { slots[i] = Slot{dev, st, nullptr, 0, true, false, nullptr}; sl = &slots[i]; }
  if (!sl) return nullptr;
  if (floats > sl->cap) {
    if (floats > ((size_t)256 << 20) / 4) return nullptr;       // larger outputs keep the atomic path
    // a stream that is being captured cannot allocate (hipMalloc would fail the capture): the launch keeps the atomic path
    // (crn_splitk_reserve sizes the scratch BEFORE a capture, so this only catches a launch that needs more
    // than any eager run before it did)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    if (sl->buf) {
      if (sl->pinned) {
        if (!sl->retired) sl->retired = new std::vector<float*>();
        sl->retired->push_back(sl->buf);                       // a captured graph may still read / write it
      } else {
        (void)hipFree(sl->buf);
      }
    }
    sl->cap = sl->pinned ? floats : std::max(floats, ((size_t)16 << 20) / 4);     // (a capture stream gets what was asked for)
    if (hipMalloc(&sl->buf, sl->cap * 4) != hipSuccess) { sl->buf = nullptr; sl->cap = 0; }
  }
  return sl->buf;
}

// arrival counters of the fused split-K reduction (mode 4): zero at allocation, every call leaves them zero
int* splitk_counters(size_t n) {
  constexpr int kMaxDev = 16;
  constexpr size_t kCap = 1 << 16;
  static int* buf[kMaxDev] = {};
  int dev = 0;
  if (n > kCap || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  if (!buf[dev]) {
    if (hipMalloc(&buf[dev], kCap * sizeof(int)) != hipSuccess) { buf[dev] = nullptr; return nullptr; }
    if (hipMemset(buf[dev], 0, kCap * sizeof(int)) != hipSuccess) return nullptr;
  }
  return buf[dev];
}

int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
unsigned magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }
// staging slots (planes padded to a power of two) needed for nplanes planes of `plane` elements
int stage_passes(int nplanes, int plane) {
  if (plane > 512) return 1 << 30;      // staging handles planes padded to at most 512 slots
  return crn_cdiv((int64_t)nplanes << ilog2_ceil(plane), 256);
}

int pad16mod32(int v) {  // smallest v' >= v with v' % 32 == 16
  int r = v % 32;
  return r <= 16 ? v + (16 - r) : v + (48 - r);
}

constexpr size_t kLdsBudget = 72 * 1024;   // 2 workgroups per CU (160 KiB LDS)

// ---------------- forward configuration search -------------------------------
struct FwdCfg {
  int MSUB, NSUB, CC, mw, mh, tsd, tsh, tsw;
  int64_t blocks;
  size_t lds;
};

// Sub-tile = 1 x mh x mw output positions (16 MFMA rows); tile = tsd x tsh x tsw sub-tiles.
// lead >= 0: vector staging with `lead` extra patch columns on the left (units of `unit` = 4 positions for
// unit-stride views, 2 for stride-2 space-to-depth views); lead < 0: scalar staging
thread_local int g_stage_unit = 4;
int patch_width(int TW, int kw, int lead) {
  return lead < 0 ? TW + kw - 1 : (lead + TW + kw - 1 + g_stage_unit - 1) / g_stage_unit * g_stage_unit;
}

bool fwd_cfg(int MSUB, int NSUB, int B, int Cin, int Npad, int D, int H, int W, int kd, int kh, int kw, int lead,
             FwdCfg* out) {
  FwdCfg c;
  c.MSUB = MSUB; c.NSUB = NSUB;
  c.mw = W >= 16 ? 16 : (W >= 8 ? 8 : (W >= 4 ? 4 : (W >= 2 ? 2 : 1)));
  c.mh = 16 / c.mw;
  const int want = 4 * MSUB;
  const int T = kd * kh * kw;
  double best = 1e300;
  c.tsd = 1; c.tsh = 1; c.tsw = want;
  for (int a = 1; a <= want; a *= 2)
    for (int bq = 1; a * bq <= want; bq *= 2) {
      const int cw = want / (a * bq);
      const int TD = a, TH = bq * c.mh, TW = cw * c.mw;
      const double tiles = (double)crn_cdiv(D, TD) * crn_cdiv(H, TH) * crn_cdiv(W, TW);
      const double patch = (double)(TD + kd - 1) * (TH + kh - 1) * patch_width(TW, kw, lead);
      const double cost = tiles * (patch + 0.25 * TD * TH * TW);   // loads + wasted MFMA rows
      if (cost < best) { best = cost; c.tsd = a; c.tsh = bq; c.tsw = cw; }
    }
  const int TD = c.tsd, TH = c.tsh * c.mh, TW = c.tsw * c.mw;
  if (lead >= 0 && (TW % g_stage_unit)) return false;
  const int PDp = TD + kd - 1, plane = (TH + kh - 1) * patch_width(TW, kw, lead);
  const int PS = PDp * plane;
  const int PSP = pad16mod32(PS), WSP = pad16mod32(T * NSUB * 16);
  auto fits = [&](int cc) {
    const bool staged = lead >= 0 ? (int64_t)cc * PS / g_stage_unit <= 256 * NVX : stage_passes(cc * PDp, plane) <= PREG;
    return (size_t)cc * (PSP + WSP) * 4 + 2 * kChTab * 4 <= kLdsBudget && staged &&
           (int64_t)cc * T * NSUB * 16 <= 256 * WREG * 4;
  };
  if (!fits(4)) return false;
  int CC = 4;
  const int cin4 = (Cin + 3) & ~3;
  while (CC * 2 <= cin4 && CC * 2 <= 64 && fits(CC * 2)) CC *= 2;
  c.CC = CC;
  c.lds = (size_t)CC * (PSP + WSP) * 4 + 2 * kChTab * 4;
  c.blocks = (int64_t)B * crn_cdiv(D, TD) * crn_cdiv(H, TH) * crn_cdiv(W, TW) * crn_cdiv(Npad, NSUB * 16);
  *out = c;
  return true;
}

}  // namespace

CrnSplitPending& crn_splitk_pending() {
  static thread_local CrnSplitPending p;
  return p;
}
bool crn_splitk_take_armed() {
  CrnSplitPending& p = crn_splitk_pending();
  const bool a = p.armed;
  p.armed = false;
  return a;
}
int crn_splitk_reduce_view(const crnView& y, const float* scratch, int splits, int accumulate, hipStream_t st) {
  return crn_splitk_reduce(y, scratch, splits, accumulate, st);
}
int crn_splitk_flush(hipStream_t st) {
  CrnSplitPending& p = crn_splitk_pending();
  if (!p.active) return CRN_OK;
  p.active = false;
  (void)st;                       // the sum is ordered behind the convolution: it runs on the conv's own stream / device
  int cur = -1;
  CRN_HIP(hipGetDevice(&cur));
  if (cur != p.device) CRN_HIP(hipSetDevice(p.device));
  const int rc = crn_splitk_reduce(p.y, p.scratch, p.splits, 0, p.stream);
  if (cur != p.device) CRN_HIP(hipSetDevice(cur));
  return rc;
}
void crn_splitk_set_pending(const crnView& y, const float* scratch, int splits, hipStream_t st) {
  CrnSplitPending& p = crn_splitk_pending();
  p.active = true; p.y = y; p.scratch = scratch; p.splits = splits; p.stream = st;
  if (hipGetDevice(&p.device) != hipSuccess) p.device = 0;
}
extern "C" int crn_splitk_defer(int on) {
  crn_splitk_pending().armed = on != 0;
  return CRN_OK;
}
float* crn_splitk_scratch(size_t floats, hipStream_t st) { return splitk_scratch(floats, st); }
// A stream that is about to be CAPTURED into a HIP graph cannot allocate: give it, ahead of the capture, a scratch as
// large as the largest one any stream of this device has needed so far (floats == 0) or `floats`.
extern "C" int crn_splitk_reserve(int64_t floats, crnStream stream) {
  int dev = 0;
  CRN_HIP(hipGetDevice(&dev));
  size_t want = floats > 0 ? (size_t)floats : 0;
  if (!want) {
    std::lock_guard<std::mutex> lock(g_sk_mu);
    for (int i = 0; i < kSkSlots; ++i)
      if (g_sk_slots[i].used && g_sk_slots[i].dev == dev) want = std::max(want, g_sk_slots[i].cap);
  }
  {
    std::lock_guard<std::mutex> lock(g_sk_mu);               // mark (or create) the stream's slot as pinned first
    SkSlot* sl = nullptr;
    for (int i = 0; i < kSkSlots && !sl; ++i)
      if (g_sk_slots[i].used && g_sk_slots[i].dev == dev && g_sk_slots[i].st == (hipStream_t)stream) sl = &g_sk_slots[i];
    for (int i = 0; i < kSkSlots && !sl; ++i)
      if (!g_sk_slots[i].used) { g_sk_slots[i] = SkSlot{dev, (hipStream_t)stream, nullptr, 0, true, false, nullptr}; sl = &g_sk_slots[i]; }
    if (!sl) return CRN_ENOMEM;
    sl->pinned = true;
  }
  if (!want) return CRN_OK;
  return splitk_scratch(want, (hipStream_t)stream) ? CRN_OK : CRN_ENOMEM;
}
// Gives back the scratch of a stream that is about to be destroyed (a capture stream whose graphs are gone); hipFree waits for
// the device, so nothing that still reads the buffer can be running.  Unknown streams are not an error.
extern "C" int crn_splitk_release(crnStream stream) {
  int dev = 0;
  CRN_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_sk_mu);
  for (int i = 0; i < kSkSlots; ++i) {
    SkSlot& sl = g_sk_slots[i];
    if (sl.used && sl.dev == dev && sl.st == (hipStream_t)stream) {
      if (sl.buf) CRN_HIP(hipFree(sl.buf));
      if (sl.retired) {
        for (float* b : *sl.retired) (void)hipFree(b);
        delete sl.retired;
      }
      sl = SkSlot{};
    }
  }
  return CRN_OK;
}
int* crn_splitk_counters(size_t n) { return splitk_counters(n); }
int crn_splitk_reduce(const crnView& y, const float* scratch, int splits, int accumulate, hipStream_t st) {
  const int64_t ytot = (int64_t)y.B * y.C * y.D * y.H * y.W;
  const int64_t per_b = (int64_t)y.C * y.D * y.H * y.W;
  if (plain_view(y) && (per_b & 3) == 0 && (((uintptr_t)scratch) & 15) == 0) {
    const int64_t total4 = ytot / 4;
    hipLaunchKernelGGL(splitk_reduce_plain_kernel, dim3((unsigned)std::min<int64_t>(crn_cdiv(total4, 256), 4096)), dim3(256),
                       0, st, y.base, y.sB, scratch, per_b / 4, total4, splits, accumulate);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(crn_cdiv(ytot, 256), 4096)), dim3(256), 0, st,
                     y, scratch, splits, accumulate);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

static long long* g_pw_stamps = nullptr;
#ifdef CRN_TOOLS      // tools/_build/libcorenet_hip_tools.so only (corenet_amd.build.build_tools)
extern "C" int crn_pw_debug_stamps(long long* out32) {       // tuning aid (CRN_PW_STAMPS=1)
  if (!g_pw_stamps) return CRN_EINVAL;
  CRN_HIP(hipDeviceSynchronize());
  CRN_HIP(hipMemcpy(out32, g_pw_stamps, 32 * sizeof(long long), hipMemcpyDeviceToHost));
  return CRN_OK;
}
#endif
extern "C" int crn_conv_fwd(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                            const float* bias, int bias_sB, const crnView* y,
                            int kd, int kh, int kw, int pd, int ph, int pw,
                            int splits, int accumulate, const crnTapBoxes* boxes, crnStream stream) {
  if (boxes && (boxes->n_groups < 0 || boxes->n_groups > 8 || boxes->c_groups < 0 || boxes->c_groups > 8)) return CRN_EINVAL;
  if (!x || !y || !w || Npad <= 0 || (Npad & 15) || x->B != y->B || kd < 1 || kh < 1 || kw < 1)
    return CRN_EINVAL;
  if (y->C > Npad) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool armed = crn_splitk_take_armed();
  { const int rcf = crn_splitk_flush(st); if (rcf != CRN_OK) return rcf; }
  const int64_t Sx = (int64_t)x->D * x->H * x->W;
  if (kd * kh * kw == 1 && pd == 0 && ph == 0 && pw == 0 && splits <= 1 && one_position(*x) && one_position(*y) &&
      x->B <= kDenseB) {
    const crnInTransform trv = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
    if (y->C >= 1024 && x->C <= 2048) {              // wide output: a thread per column
      hipLaunchKernelGGL(dense_cols_kernel, dim3((unsigned)crn_cdiv(y->C, 64)), dim3(256), (size_t)x->B * x->C * 4, st,
                         x->base, x->sB, x->sC, trv, w, Npad, y->C, x->C, x->B, bias, bias_sB, y->base, y->sB, y->sC,
                         accumulate);
      CRN_CHECK_LAUNCH();
      return CRN_OK;
    }
    if (y->C <= 256 && !trv.scale && x->C >= 1024) {   // long reduction onto few columns
      const int zs = (y->sC == 1 && y->sB >= y->C && !crn_deterministic()) ? std::max(1, std::min(16, x->C / 2048)) : 1;   // slices of >= 2048 rows
      if (zs > 1 && !accumulate)
        CRN_HIP(hipMemset2DAsync(y->base, (size_t)y->sB * 4, 0, (size_t)y->C * 4, (size_t)x->B, st));
      hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)crn_cdiv(y->C, 8), (unsigned)x->B, (unsigned)zs), dim3(256), 0, st,
                         x->base, x->sB, x->sC, w, Npad, y->C, x->C, bias, bias_sB, y->base, y->sB, y->sC, accumulate,
                         crn_cdiv(x->C, zs));
      CRN_CHECK_LAUNCH();
      return CRN_OK;
    }
  }
  if (kd * kh * kw == 1 && pd == 0 && ph == 0 && pw == 0 && splits <= 1 && plain_view(*x) && flat_out_view(*y) &&
      Sx == (int64_t)y->D * y->H * y->W && (Sx & 3) == 0 && (((uintptr_t)w) & 15) == 0 &&
      (int64_t)x->C * Sx * 4 < ((int64_t)1 << 31) && (int64_t)x->C * Npad * 4 < ((int64_t)1 << 31) &&
      (!tr || !tr->scale || x->C <= kPwTab)) {           // (buffer offsets of the hand-tracked loads; the LDS tables)
    static const bool pw2_on = getenv("CRN_PW2") && atoi(getenv("CRN_PW2")) != 0;   // measured (tools/pw_trace.sh, profiles/r03_pw_trace.txt): 13-15 us per layer like the staged kernel -- both sit at ~38 TF/s on these 0.5 GFLOP launches -- plus a split-K reduction where it splits: off by default
    if (pw2_on && plain_view(*y) && (x->C & 15) == 0 && x->C >= 64 && (y->sB & 3) == 0 && (y->sC & 3) == 0) {
      // operands straight from HBM (pw2_kernel).  Decomposition: 64 x 64 tiles; 8 waves per workgroup (4 for 64 input
      // channels), one or two 16-channel groups per wave; K slices of NW * GPW * 16 channels over blockIdx.z
      Pw2Geom q{};
      q.x = x->base; q.w = w; q.bias = bias; q.y = y->base;
      q.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
      q.B = x->B; q.C = x->C; q.N = y->C; q.Npad = Npad; q.S = (int)Sx; q.tilesS = (int)crn_cdiv(Sx, 64);
      q.xsB = x->sB; q.ysB = y->sB; q.ysC = y->sC; q.bias_sB = bias_sB; q.mode = accumulate ? 1 : 0;
      const int groups = x->C / 16;
      const int NW = groups >= 8 ? 8 : 4;
      const int64_t tiles = (int64_t)q.tilesS * x->B * crn_cdiv(y->C, 64);
      static const int pw2_fill = getenv("CRN_PW2_FILL") ? atoi(getenv("CRN_PW2_FILL")) : 384;
      int GPW = (groups >= 2 * NW && tiles * crn_cdiv(groups, 2 * NW) >= pw2_fill) ? 2 : 1;
      if (const char* f = getenv("CRN_PW2_GPW")) GPW = atoi(f) == 2 ? 2 : 1;
      q.ksl = NW * GPW * 16;
      q.ksplits = crn_cdiv(x->C, q.ksl);
      float* scratch2 = nullptr;
      if (q.ksplits > 1) {
        const int64_t ytot2 = (int64_t)y->B * y->C * Sx;
        scratch2 = splitk_scratch((size_t)q.ksplits * ytot2, st);
        if (scratch2) { q.y = scratch2; q.ysB = (int64_t)y->C * Sx; q.ysC = Sx; q.mode = 0; }
      }
      if (q.ksplits == 1 || scratch2) {
        const dim3 grid((unsigned)(q.tilesS * x->B), (unsigned)crn_cdiv(y->C, 64), (unsigned)q.ksplits);
        int rc2;
        if (NW == 8) rc2 = GPW == 2 ? launch_pw2<8, 2>(q, grid, st) : launch_pw2<8, 1>(q, grid, st);
        else rc2 = launch_pw2<4, 1>(q, grid, st);
        if (rc2 != CRN_OK) return rc2;
        if (q.ksplits > 1) {
          if (armed && !accumulate && y->sB == (int64_t)y->C * Sx) {
            crn_splitk_set_pending(*y, scratch2, q.ksplits, st);
            return CRN_OK;
          }
          return crn_splitk_reduce(*y, scratch2, q.ksplits, accumulate, st);
        }
        return CRN_OK;
      }
    }
    PwGeom p{};
    p.x = x->base; p.y = y->base; p.w = w; p.bias = bias;
    p.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
    p.B = x->B; p.C = x->C; p.N = y->C; p.Npad = Npad; p.S = (int)Sx;
    p.xsB = x->sB; p.ysB = y->sB; p.bias_sB = bias_sB; p.mode = accumulate ? 1 : 0;
    p.ysC = y->sC; p.ysP = y->sW;
    p.splits = 1; p.cps = (x->C + 63) & ~63;
    // long reductions over few positions (stage4/5 of the encoder: K = 1024..2048, 64..256 positions per sample):
    // split the channels over blocks, partial sums to the split-K scratch, one reduction launch
    // 64-column tiles (CRN_PW_NS=4) were measured equal to 32-column ones on the whole step (10.07 vs 10.04 ms): these
    // launches are latency bound, not operand-traffic bound; 32 stays the default
    static const char* pw_ns_env = getenv("CRN_PW_NS");
    const bool wide = pw_ns_env && atoi(pw_ns_env) == 4 && Npad % 64 == 0;
    const int BNh = wide ? 64 : 32;
    const int64_t blocks = crn_cdiv(Sx, 64) * crn_cdiv(y->C, BNh) * (int64_t)x->B;
    const int64_t ytot = (int64_t)y->B * y->C * Sx;
    static const bool pw_nosplit = getenv("CRN_PW_NOSPLIT") != nullptr;
    float* scratch = nullptr;
    static const int pw_blocks = getenv("CRN_PW_BLOCKS") ? atoi(getenv("CRN_PW_BLOCKS")) : 256;   // tuning aids
    static const int pw_fill = getenv("CRN_PW_FILL") ? atoi(getenv("CRN_PW_FILL")) : 512;
    static const int pw_minc = getenv("CRN_PW_MINC") ? atoi(getenv("CRN_PW_MINC")) : 128;
    if (splits < 1 && !pw_nosplit && blocks < pw_blocks && x->C >= 2 * pw_minc) {
      int sp = (int)std::min<int64_t>(std::min<int64_t>(x->C / pw_minc, 16), crn_cdiv(pw_fill, blocks));
      if (sp > 1) {
        p.cps = (crn_cdiv(x->C, sp) + 31) & ~31;
        sp = crn_cdiv(x->C, p.cps);
        if (sp > 1 && (scratch = splitk_scratch((size_t)sp * ytot, st)) != nullptr) {
          p.yr = y->base; p.yr_sB = y->sB; p.yr_sC = y->sC; p.yr_sP = y->sW; p.yr_accumulate = accumulate ? 1 : 0;
          p.splits = sp; p.y = scratch; p.ysB = (int64_t)y->C * Sx; p.ysC = Sx; p.ysP = 1; p.mode = 0;
          static const bool sk_launch_pw = getenv("CRN_SPLITK_FUSED") == nullptr;
          if (!sk_launch_pw) p.counters = splitk_counters((size_t)crn_cdiv(Sx, 64) * crn_cdiv(y->C, BNh) * x->B);
        } else {
          p.cps = (x->C + 31) & ~31;
        }
      }
    }
    static const int pw_ablate = getenv("CRN_PW_ABLATE") ? atoi(getenv("CRN_PW_ABLATE")) : 0;
    p.ablate = pw_ablate;
#ifdef CRN_TOOLS
    static const bool want_stamps = getenv("CRN_PW_STAMPS") != nullptr;
#else
    constexpr bool want_stamps = false;
#endif
    static long long* pw_stamps = nullptr;
    if (want_stamps) {
      if (!pw_stamps) CRN_HIP(hipMalloc(&pw_stamps, 32 * sizeof(long long)));
      CRN_HIP(hipMemsetAsync(pw_stamps, 0, 32 * sizeof(long long), st));
      p.stamps = pw_stamps;
      g_pw_stamps = pw_stamps;
    }
    dim3 grid((unsigned)crn_cdiv(Sx, 64), (unsigned)crn_cdiv(y->C, BNh), (unsigned)(x->B * p.splits));
    const size_t tab_bytes = p.tr.scale ? (size_t)2 * p.cps * sizeof(float) : 0;
    if (wide) hipLaunchKernelGGL(pointwise_fwd_kernel<4>, grid, dim3(256), tab_bytes, st, p);
    else hipLaunchKernelGGL(pointwise_fwd_kernel<2>, grid, dim3(256), tab_bytes, st, p);
    CRN_CHECK_LAUNCH();
    if (p.splits > 1 && !p.counters) {
      if (armed && !accumulate && plain_view(*y) && y->sB == (int64_t)y->C * Sx) {
        crn_splitk_set_pending(*y, scratch, p.splits, st);
        return CRN_OK;
      }
      return crn_splitk_reduce(*y, scratch, p.splits, accumulate, st);
    }
    return CRN_OK;
  }
  // Score every (MSUB, NSUB) tile: useful MFMA rows x operand reuse of the tile x how well
  // the grid (with split-K as a fallback) fills 256 CUs.
  static const int kM[4] = {8, 4, 2, 1};
  static const int kN[3] = {4, 2, 1};
  static const bool no_vec = getenv("CRN_NO_VEC") != nullptr;
  // staging of x: float4 units (unit-stride views), position pairs (stride-2 space-to-depth views), scalars
  const int xmode = no_vec ? 0 : (vec_view(*x) ? 1 : ((x->sW == 2 && (x->W & 1) == 0) ? 2 : 0));   // s2d views and stride-2 views
  g_stage_unit = xmode == 2 ? 2 : 4;
  const int lead = xmode == 1 ? vec_lead(pw) : (xmode == 2 ? ((pw % 2) + 2) % 2 : -1);
  // Estimated cycles per CU (calibrated on tools/sweep_fwd.sh): work that adds up on the SIMDs
  // (MFMA issue at ~77% + staging instructions, which do not hide under MFMAs) plus one exposed
  // load->commit->barrier latency per chunk and pair of resident workgroups; split-K pays a
  // zero-fill launch and an atomic epilogue.
  // split-K now costs one small reduction launch (tuned on bench.py: 20000 / 512 beat 40000 / 256 by 0.3 ms)
  static const double kSplitPenalty = getenv("CRN_SPLIT_PENALTY") ? atof(getenv("CRN_SPLIT_PENALTY")) : 20000.0;
  static const int kSplitFill = getenv("CRN_SPLIT_FILL") ? atoi(getenv("CRN_SPLIT_FILL")) : 512;
  FwdCfg best{}; bool have = false; double best_cost = 1e300; int best_splits = 1;
  for (int mi = 0; mi < 4; ++mi)
    for (int ni = 0; ni < 3; ++ni) {
      if (kN[ni] > 1 && kN[ni] * 16 > Npad) continue;
      if (kM[mi] * kN[ni] > 8) continue;              // instantiated tiles (conv_kernels.h)
      FwdCfg c;
      if (!fwd_cfg(kM[mi], kN[ni], y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, lead, &c)) continue;
      const int T = kd * kh * kw, nchunks = crn_cdiv(x->C, c.CC), NB = kN[ni] * 16;
      const int TD = c.tsd, TH = c.tsh * c.mh, TW = c.tsw * c.mw;
      const double patch = (double)c.CC * (TD + kd - 1) * (TH + kh - 1) * patch_width(TW, kw, lead);
      const double slots = (lead >= 0 ? patch / g_stage_unit / 256 * (g_stage_unit == 4 ? 450.0 : 400.0) : patch / 256 * 350.0) +
                           (double)c.CC * T * NB / 4 / 256 * 300.0;
      const double chunk_work = (double)kM[mi] * kN[ni] * (c.CC / 4) * T * 32.0 * 1.3 + slots;
      for (int sp = 1; sp <= (splits < 1 ? std::min(nchunks, 64) : 1); sp *= 2) {
        if (sp > 1 && c.blocks * (sp / 2) >= kSplitFill) break;   // split only to fill the chip
        const int cps = crn_cdiv(nchunks, sp);
        const double bpc = std::max(1.0, std::ceil((double)c.blocks * sp / 256.0));
        double cost = bpc * cps * chunk_work + cps * std::ceil(bpc / 2.0) * 9000.0 + (sp > 1 ? kSplitPenalty : 0.0);
        if (cost < best_cost) { best_cost = cost; best = c; have = true; best_splits = sp; }
      }
    }
  bool forced = false;
  if (const char* f = getenv("CRN_FWD_FORCE")) {     // tuning aid: "MSUB,NSUB"
    int fM, fN; FwdCfg c;
    if (sscanf(f, "%d,%d", &fM, &fN) == 2 && fwd_cfg(fM, fN, y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, lead, &c)) {
      best = c; have = true; forced = true;
    }
  }
  int xvec = lead >= 0 ? xmode : 0;
  if (!have && xvec) {       // no 16-byte configuration fits (tiny W): scalar staging
    xvec = 0;
    for (int mi = 0; mi < 4 && !have; ++mi) {
      FwdCfg c;
      if (fwd_cfg(kM[mi], 1, y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, -1, &c)) { best = c; have = true; forced = true; }
    }
  }
  if (!have) return CRN_EINVAL;
  ConvGeom g{};
  g.x = *x; g.y = *y;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.w = w; g.bias = bias; g.Npad = Npad; g.bias_sB = bias_sB;
  g.lead = xvec ? lead : 0;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw + g.lead; g.T = kd * kh * kw;
  g.mw = best.mw; g.mh = best.mh; g.nsh = best.tsh; g.nsw = best.tsw;
  g.TD = best.tsd; g.TH = best.tsh * best.mh; g.TW = best.tsw * best.mw;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = patch_width(g.TW, kw, xvec ? lead : -1);
  g.PS = g.PD * g.PH * g.PW; g.PSP = pad16mod32(g.PS);
  g.tilesD = crn_cdiv(y->D, g.TD); g.tilesH = crn_cdiv(y->H, g.TH); g.tilesW = crn_cdiv(y->W, g.TW);
  const int NSUB = best.NSUB, CC = best.CC;
  g.CC = CC; g.WSP = pad16mod32(g.T * NSUB * 16);
  g.nchunks = crn_cdiv(x->C, CC);
  if (splits < 1) {   // auto split-K (atomic accumulate) only when the output grid cannot fill the chip
    static const char* fs = getenv("CRN_FWD_SPLITS");
    splits = fs ? atoi(fs) : (forced ? (best.blocks >= 192 ? 1 : (int)std::min<int64_t>(std::min(g.nchunks, 16), crn_cdiv(256, best.blocks)))
                                     : best_splits);
  }
  if (splits > g.nchunks) splits = g.nchunks;
  g.chunks_per_split = crn_cdiv(g.nchunks, splits);
  splits = crn_cdiv(g.nchunks, g.chunks_per_split);
  g.mode = splits > 1 ? 2 : (accumulate ? 1 : 0);
  g.lg2 = ilog2_ceil(g.PH * g.PW); g.npass = xvec ? 0 : stage_passes(CC * g.PD, g.PH * g.PW);
  g.magic_PW = magic20(g.PW); g.magic_PD = magic20(g.PD); g.magic_T = magic20(g.T);
  if (xvec) {
    g.plu = g.PH * g.PW / g_stage_unit; g.pw4 = g.PW / g_stage_unit; g.nunits = CC * g.PD * g.plu;
    g.magic_PLU = magic20(g.plu); g.magic_PW4 = magic20(g.pw4);
  }
  // split-K: partial sums into a dense scratch tensor + one reduction launch (mode 3); atomics (mode 2)
  // only when the scratch cannot be had
  const crnView yreal = *y;
  const int64_t ytot = (int64_t)y->B * y->C * y->D * y->H * y->W;
  static const bool sk_atomic = getenv("CRN_SPLITK_ATOMIC") != nullptr;
  float* scratch = (g.mode == 2 && !sk_atomic) ? splitk_scratch((size_t)splits * ytot, st) : nullptr;
  static const bool sk_launch = getenv("CRN_SPLITK_FUSED") == nullptr;   // default: the separate reduction launch (see conv_kernels.h, mode 4)
  if (scratch) {
    g.mode = 3;
    g.yreal = *y; g.accumulate_real = accumulate ? 1 : 0;
    const int64_t S = (int64_t)y->D * y->H * y->W;
    g.y.base = scratch; g.y.chan_off = nullptr;
    g.y.sW = 1; g.y.sH = y->W; g.y.sD = y->H * y->W; g.y.sC = S; g.y.sB = (int64_t)y->C * S;
  }
  y = &g.y;                                // epilogue alignment below refers to the tensor actually written
  if (g.mode == 2 && !accumulate) {
    hipLaunchKernelGGL(zero_view_kernel, dim3((unsigned)std::min<int64_t>(crn_cdiv(ytot, 256), 4096)),
                       dim3(256), 0, st, yreal);
    CRN_CHECK_LAUNCH();
  }
  dim3 grid((unsigned)(g.tilesD * g.tilesH * g.tilesW * y->B), (unsigned)crn_cdiv(Npad, NSUB * 16),
            (unsigned)splits);
  if (g.mode == 3 && !sk_launch && (g.counters = splitk_counters((size_t)grid.x * grid.y)) != nullptr) g.mode = 4;
  const size_t lds_bytes = best.lds;
  {
    bool al = y->sW == 1 && g.mw >= 4 && (y->W & 3) == 0 && (y->sH & 3) == 0 && (y->sD & 3) == 0 && (y->sB & 3) == 0 &&
              (y->sC & 3) == 0 && (((uintptr_t)y->base) & 15) == 0 && y->chan_off == nullptr && g.mode != 2;
    g.vec_store = al ? 1 : 0;
  }
  static const bool no_boxes = getenv("CRN_NO_BOXES") != nullptr;
  if (boxes && !no_boxes) {
    // groups must divide the channel counts, otherwise the information is ignored
    if (boxes->n_groups > 0 && y->C % boxes->n_groups == 0) { g.n_groups = boxes->n_groups; memcpy(g.n_box, boxes->n_box, sizeof(g.n_box)); }
    if (boxes->c_groups > 0 && x->C % boxes->c_groups == 0) { g.c_groups = boxes->c_groups; memcpy(g.c_box, boxes->c_box, sizeof(g.c_box)); }
  }
  g.dbg = getenv("CRN_DBG_MODE") ? atoi(getenv("CRN_DBG_MODE")) : 0;
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_fwd] x(C%d %dx%dx%d) y(C%d %dx%dx%d) k%dx%dx%d: MSUB %d NSUB %d CC %d tile %dx%dx%d "
            "grid %ux%ux%u lds %zu npass %d lg2 %d xvec %d units %d\n", x->C, x->D, x->H, x->W, y->C, y->D, y->H, y->W, kd, kh, kw,
            best.MSUB, NSUB, CC, g.TD, g.TH, g.TW, grid.x, grid.y, grid.z, lds_bytes, g.npass, g.lg2, xvec, g.nunits);
  int rc = CRN_EINVAL;
#define CRN_FWD_CASE(M, N) if (best.MSUB == M && NSUB == N) rc = crn_launch_fwd_##M##_##N(g, xvec, grid, lds_bytes, st);
  CRN_FWD_CONFIGS(CRN_FWD_CASE)
#undef CRN_FWD_CASE
  if (rc == CRN_OK && g.mode == 3) {
    // the next BatchRenorm launch over y adds the partial sums up itself when the caller armed it (crn_splitk_defer) -- like the
    // pointwise and the split-bf16 paths above: the encoder's fp32 3x3 layers and decoder stage 2 lose their reduction launch
    if (armed && !accumulate && plain_view(yreal) && yreal.sB == (int64_t)yreal.C * yreal.D * yreal.H * yreal.W) {
      crn_splitk_set_pending(yreal, scratch, splits, st);
      return rc;
    }
    rc = crn_splitk_reduce(yreal, scratch, splits, accumulate, st);
  }
  return rc;
}

namespace {
struct WgCand { int TD, TH, TW, RSUB, NSUB, CC, PW, PSP, POSP; double cost; size_t lds; };

// One weight-grad configuration: position tile TDxTHxTW, RSUB x NSUB MFMA tiles per wave.
// xlead >= 0: x staged in 16-byte units; dvec: dy staging mode (0 scalar, 1 float4 units, 2 position pairs).
bool wg_cand(int TD, int TH, int TW, int RSUB, int NSUB, int Cin, int Npad, int kd, int kh, int kw, int xlead,
             int dvec, size_t lds_budget, WgCand* out) {
  const int T = kd * kh * kw, NB = NSUB * 16, npos = TD * TH * TW;
  if (npos > 512 || RSUB * NSUB > 8 || (NSUB > 1 && NB > Npad)) return false;
  if ((xlead >= 0 || dvec) && (TW & 3)) return false;
  if (dvec == 1 ? (int64_t)NB * npos / 4 > 256 * NV
                : dvec == 2 ? (int64_t)NB * npos / 2 > 256 * NVX : stage_passes(NB, npos) > DREG) return false;
  const int PDp = TD + kd - 1, PH = TH + kh - 1, PW = patch_width(TW, kw, xlead), plane = PH * PW;
  int CC = std::min(std::min(std::max(1, (64 * RSUB) / T), Cin), 64);
  auto staged = [&](int cc) {
    return xlead >= 0 ? (int64_t)cc * PDp * plane / 4 <= 256 * NVX : stage_passes(cc * PDp, plane) <= PREG;
  };
  while (CC > 1 && !staged(CC)) --CC;
  if (!staged(CC)) return false;
  const int PSP = xlead >= 0 ? pad16mod32(PDp * plane) : PDp * plane + 1;
  const int POSP = npos + 4;
  const size_t lds = ((size_t)CC * PSP + (size_t)NB * POSP + 2 * kChTab) * 4;
  if (lds > lds_budget) return false;
  const int rows = CC * T;
  if (rows > 64 * RSUB) return false;
  // cycles per wave and tile: MFMA issue plus staging instructions (they do not overlap: every
  // VALU/SALU instruction costs MFMA issue time), per useful multiply-add
  const double mfma = (double)RSUB * NSUB * (npos / 4.0) * 32.0;
  const double xslots = xlead >= 0 ? (double)CC * PDp * plane / 4 / 256 * 450.0 : (double)CC * PDp * plane / 256 * 350.0;
  const double dslots = dvec == 1 ? (double)NB * npos / 4 / 256 * 300.0
                                  : dvec == 2 ? (double)NB * npos / 2 / 256 * 300.0 : (double)NB * npos / 256 * 300.0;
  const double useful = (double)rows * std::min(NB, Npad) * npos;
  *out = WgCand{TD, TH, TW, RSUB, NSUB, CC, PW, PSP, POSP, (mfma + xslots + dslots + 800.0) / useful, lds};
  return true;
}
}  // namespace

namespace {
// One weight-gradient launch: window kd x kh x kw, which may be the sub-box (bd0,bh0,bw0) of a packed
// window khf x kwf with Tfull taps; columns [0, ncols) of dw (row stride Npad).
int wgrad_launch(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad, int ncols,
                 int kd, int kh, int kw, int pd, int ph, int pw, int Tfull, int khf, int kwf, int bd0, int bh0,
                 int bw0, int max_blocks, const crnTapBoxes* boxes, hipStream_t st) {
  const int T = kd * kh * kw;
  g_stage_unit = 4;                              // x patches of the weight gradient: float4 units only
  const int NpadC = (ncols + 15) & ~15;          // columns this launch covers
  const int Dy = dy->D, Hy = dy->H, Wy = dy->W;
  const int TWc = Wy >= 16 ? 16 : ((Wy + 3) & ~3);
  static const bool no_vec = getenv("CRN_NO_VEC") != nullptr;
  int xlead = (vec_view(*x) && !no_vec) ? vec_lead(pw) : -1;
  // dy staging: float4 units for unit-stride views, position pairs for the stride-2 space-to-depth views of
  // the transposed convolutions (one float4 holds two positions of this parity), scalars otherwise
  int dvec = no_vec ? 0 : (vec_view(*dy) ? 1 : ((dy->sW == 2 && dy->chan_off != nullptr && (dy->W & 1) == 0) ? 2 : 0));
  WgCand best{}; bool have = false;
  auto consider = [&](int TD, int TH, int RSUB, int NSUB, size_t budget) {
    WgCand c;
    if (!wg_cand(std::min(TD, Dy), std::min(TH, Hy), TWc, RSUB, NSUB, x->C, NpadC, kd, kh, kw, xlead, dvec, budget, &c))
      return false;
    if (RSUB > 1 && c.CC * T <= 32 * RSUB) return false;   // a smaller RSUB covers it
    if (!have || c.cost < best.cost) { best = c; have = true; }
    return true;
  };
  auto search = [&]() {
    static const int kTD[4] = {4, 3, 2, 1};
    static const int kTH[6] = {16, 8, 6, 4, 2, 1};
    for (int tdi = 0; tdi < 4; ++tdi)
      for (int thi = 0; thi < 6; ++thi)
        for (int NSUB = 4; NSUB >= 1; NSUB >>= 1)
          for (int RSUB = 8; RSUB >= 1; RSUB >>= 1) consider(kTD[tdi], kTH[thi], RSUB, NSUB, kLdsBudget);
  };
  search();
  if (!have && (xlead >= 0 || dvec)) { xlead = -1; dvec = 0; search(); }   // tiny maps: scalar staging
  if (const char* f = getenv("CRN_WG_FORCE")) {      // tuning aid: "TD,TH,RSUB,NSUB"
    int fTD, fTH, fR, fN;
    if (sscanf(f, "%d,%d,%d,%d", &fTD, &fTH, &fR, &fN) == 4) {
      WgCand c;
      if (wg_cand(std::min(fTD, Dy), std::min(fTH, Hy), TWc, fR, fN, x->C, NpadC, kd, kh, kw, xlead, dvec, 150 * 1024, &c)) {
        best = c; have = true;
      }
    }
  }
  if (!have) return CRN_EINVAL;
  if (xlead < 0) dvec = 0;                           // instantiated variants: scalar, x, x+dy, x+dy pairs
  if (xlead < 0 || !dvec) {
    // the candidate was costed with dvec; re-derive it with the variant that will run
    WgCand c;
    if (!wg_cand(best.TD, best.TH, best.TW, best.RSUB, best.NSUB, x->C, NpadC, kd, kh, kw, xlead, dvec, 150 * 1024, &c)) {
      have = false; search();
      if (!have) return CRN_EINVAL;
    } else {
      best = c;
    }
  }
  WgradGeom g{};
  g.x = *x; g.dy = *dy;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.dw = dw; g.Npad = Npad;
  g.lead = xlead >= 0 ? xlead : 0;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw + g.lead; g.T = T;
  g.TD = best.TD; g.TH = best.TH; g.TW = best.TW;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = best.PW;
  g.PSP = best.PSP; g.POSP = best.POSP;
  const int NSUB = best.NSUB, RSUB = best.RSUB, NB = NSUB * 16, CC = best.CC;
  const int npos = g.TD * g.TH * g.TW;
  g.CC = CC;
  g.tilesD = crn_cdiv(Dy, g.TD); g.tilesH = crn_cdiv(Hy, g.TH); g.tilesW = crn_cdiv(Wy, g.TW);
  g.ntiles = g.tilesD * g.tilesH * g.tilesW * dy->B;
  const int cblocks = crn_cdiv(x->C, CC), nblocks = crn_cdiv(NpadC, NB);
  static const int kWgBlocks = getenv("CRN_WG_BLOCKS") ? atoi(getenv("CRN_WG_BLOCKS")) : 512;
  // one resident round: at most kWgBlocks (= 2 per CU) workgroups, or the stragglers double the time
  const int budget = max_blocks > 0 ? std::min(max_blocks, kWgBlocks) : kWgBlocks;
  int splits = std::max(1, std::min(g.ntiles, budget / (cblocks * nblocks)));
  if (crn_deterministic()) splits = 1;                       // (the per-group splits below follow: min(ntiles, ~splits))
  g.tiles_per_split = crn_cdiv(g.ntiles, splits);
  splits = crn_cdiv(g.ntiles, g.tiles_per_split);
  bool balanced = false;
  if (boxes && boxes->n_groups > 0 && boxes->n_groups <= 8 && dy->C % boxes->n_groups == 0 &&
      (dy->C / boxes->n_groups) % NB == 0) {           // every block's columns belong to one group
    balanced = true;
    g.n_groups = boxes->n_groups; memcpy(g.n_box, boxes->n_box, sizeof(g.n_box));
    // blocks of a group with fewer real taps do less MFMA work per tile: give them proportionally more
    // tiles (fewer splits), the heavy groups more splits; the total stays within the resident budget
    double w[8], wsum = 0.0;
    for (int gi = 0; gi < g.n_groups; ++gi) {
      const signed char* b = boxes->n_box[gi];
      w[gi] = 0.35 * T + std::max(0, b[1] - b[0]) * std::max(0, b[3] - b[2]) * std::max(0, b[5] - b[4]);  // + staging share
      wsum += w[gi];
    }
    g.balanced = 1; g.nbpg = (dy->C / boxes->n_groups) / NB;
    g.zoff[0] = 0;
    for (int gi = 0; gi < g.n_groups; ++gi) {
      const int sgi = crn_deterministic() ? 1 : std::max(1, std::min(g.ntiles, (int)(splits * w[gi] * g.n_groups / wsum + 0.5)));
      g.tps[gi] = crn_cdiv(g.ntiles, sgi);
      g.sg[gi] = crn_cdiv(g.ntiles, g.tps[gi]);
      g.zoff[gi + 1] = g.zoff[gi] + g.nbpg * g.sg[gi];
    }
  }
  g.lg2 = ilog2_ceil(g.PH * g.PW); g.npass = xlead >= 0 ? 0 : stage_passes(CC * g.PD, g.PH * g.PW);
  g.dlg2 = ilog2_ceil(npos); g.dnpass = dvec ? 0 : stage_passes(NB, npos);
  g.magic_PW = magic20(g.PW); g.magic_PD = magic20(g.PD); g.magic_T = magic20(T);
  g.magic_TW = magic20(g.TW); g.magic_TH = magic20(g.TH);
  if (xlead >= 0) {
    g.plu = g.PH * g.PW / 4; g.pw4 = g.PW / 4; g.nunits = CC * g.PD * g.plu;
    g.magic_PLU = magic20(g.plu); g.magic_PW4 = magic20(g.pw4);
  }
  if (dvec) { g.np4 = npos / (dvec == 2 ? 2 : 4); g.dnunits = NB * g.np4; g.magic_NP4 = magic20(g.np4); }
  g.Tfull = Tfull; g.khf = khf; g.kwf = kwf; g.bd0 = bd0; g.bh0 = bh0; g.bw0 = bw0; g.ncols = ncols;
  g.dbg = getenv("CRN_DBG_MODE") ? atoi(getenv("CRN_DBG_MODE")) : 0;
  // (with per-group split counts the idle tail blocks must stay spread over the XCDs: plain round robin)
  g.xcd = getenv("CRN_WG_XCD") ? atoi(getenv("CRN_WG_XCD")) : 1;
  dim3 grid((unsigned)cblocks, (unsigned)(g.balanced ? 1 : nblocks), (unsigned)(g.balanced ? g.zoff[g.n_groups] : splits));
  const size_t lds_bytes = best.lds;
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_wgrad] x(C%d %dx%dx%d) dy(C%d %dx%dx%d) k%dx%dx%d: RSUB %d NSUB %d CC %d tile %dx%dx%d "
            "grid %ux%ux%u lds %zu npass %d dnpass %d tiles/split %d xvec %d dvec %d units %d/%d\n", x->C, x->D, x->H,
            x->W, dy->C, dy->D, dy->H, dy->W, kd, kh, kw, RSUB, NSUB, CC, g.TD, g.TH, g.TW, grid.x, grid.y, grid.z,
            lds_bytes, g.npass, g.dnpass, g.tiles_per_split, xlead >= 0, (int)dvec, g.nunits, g.dnunits);
#define CRN_WG_CASE(R, N) \
  if (RSUB == R && NSUB == N) return crn_launch_wgrad_##R##_##N(g, xlead >= 0, dvec, grid, lds_bytes, st);
  CRN_WG_CONFIGS(CRN_WG_CASE)
#undef CRN_WG_CASE
  return CRN_EINVAL;
}
}  // namespace

extern "C" int crn_conv_wgrad(const crnView* x, const crnInTransform* tr, const crnView* dy,
                              float* dw, int Npad, int kd, int kh, int kw, int pd, int ph, int pw,
                              int zero_first, const crnTapBoxes* boxes, crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !dy || !dw || Npad <= 0 || (Npad & 15) || x->B != dy->B) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T = kd * kh * kw;
  if (T > 512) return CRN_EINVAL;
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * T * Npad * 4, st));
  if (T == 1 && one_position(*x) && one_position(*dy) && dy->C >= 1024) {      // dense layer on a 1^3 grid (stage_1)
    const crnInTransform trv = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)crn_cdiv(dy->C, 256), (unsigned)x->C), dim3(256), 0, st, x->base,
                       x->sB, x->sC, trv, dy->base, dy->sB, dy->sC, dw, Npad, dy->C, x->C, x->B);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  // Tap boxes: inside the single launch every block enumerates only the (channel, tap) rows of its output
  // group's box (conv_wgrad_kernel).  The alternative -- one launch per output parity with a smaller
  // window -- LOSES on MI355X (eight prologues and atomic epilogues: s5t1 1.02 ms vs 0.94 ms) and stays an
  // experiment (CRN_WG_BOXES=1).
  static const bool wg_boxes = getenv("CRN_WG_BOXES") != nullptr;
  if (boxes && wg_boxes && boxes->n_groups > 1 && boxes->n_groups <= 8 && dy->C % boxes->n_groups == 0 &&
      ((dy->C / boxes->n_groups) & 15) == 0 && dy->chan_off != nullptr) {
    const int per = dy->C / boxes->n_groups;
    for (int gi = 0; gi < boxes->n_groups; ++gi) {
      const signed char* b = boxes->n_box[gi];
      crnView dyg = *dy;
      dyg.C = per; dyg.chan_off = dy->chan_off + (size_t)gi * per;
      const int rc = wgrad_launch(x, tr, &dyg, dw + (size_t)gi * per, Npad, per, b[1] - b[0], b[3] - b[2], b[5] - b[4],
                                  pd - b[0], ph - b[2], pw - b[4], T, kh, kw, b[0], b[2], b[4], 0, nullptr, st);
      if (rc != CRN_OK) return rc;
    }
    return CRN_OK;
  }
  static const bool no_boxes = getenv("CRN_NO_BOXES") != nullptr;
  return wgrad_launch(x, tr, dy, dw, Npad, Npad, kd, kh, kw, pd, ph, pw, T, kh, kw, 0, 0, 0, 0, no_boxes ? nullptr : boxes, st);
}
