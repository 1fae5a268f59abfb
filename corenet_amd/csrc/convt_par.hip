// ConvTranspose3d(k = 7, stride 2, padding 3, output_padding 1) with 16 input channels -- decoder stage_6.t1, the
// layer that writes the logits (reconstruction_decoder.py:89-95; 2 classes for h7, 14 assumed for m7 / m9) -- on the
// split-bf16 MFMA (three v_mfma_f32_16x16x32_bf16 per fp32 product, see conv_bf3.hip), forward and data gradient.
//
// Why its own kernels.  o = 2 i - 3 + k: an output voxel o = 2 q + r of parity r = (rd, rh, rw) only sees the taps
// k = 5 - 2 z + r of the 4^3 window z around q -- 3 per dimension for r = 0, 4 for r = 1, 343 of 8 x 64 in all.  The generic
// engine (conv_bf3.hip) runs the layer as ONE window correlation onto 8 C parity channels in 16-column blocks: with C = 14
// the blocks straddle parities (tap boxes become unions, almost nothing is skipped), every one of the 7 blocks re-stages the
// same input patch, and a staging step feeds 16 MFMA triples per wave (forward 768 us, data gradient 1036 us at B = 4, 0.19-0.25
// of the engine's roof; profiles/r06_layer_times_bf16x3_c14.txt).  Here a workgroup keeps the patch of its 4 x 8 x 16 tile
// -- both 8-channel chunks, 98 KB -- in LDS for the whole tile and WALKS the parities: for each (rd, rh) pair it multiplies
// exactly the (3 + rd) x (3 + rh) window rows that hold weights, the two rw parities side by side as two 16-column blocks
// (one A fragment pair feeds both), and stores 8 consecutive floats per lane (rw interleaved): 196 MFMA triples per sub-tile
// instead of 224 + re-staging, weights streamed through a double-buffered LDS slab by LDS-DMA (no registers, one barrier per
// step).  The data gradient is the same walk over the space-to-depth channels (rd, rh, rw, n) of dy with parity-pure chunks,
// both rw of a chunk staged from ONE 16-byte load per (row, 2 positions, channel).
//
// Weights arrive as an IMAGE in step order, built by crn_bf3_gather_image from the flat parameter slab with a host-made
// index table (conv_geometry.convt_par_*_table): per step hi[rows][4 taps][32 columns] then lo[...], entries of 8 bf16.
#include "conv_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace crnk;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kT = 512;                       // 8 waves: 32 sub-tiles of 16 W positions, 4 per wave
constexpr int TD = 4, TH = 8, TW = 16;        // tile (positions of the coarse grid)
constexpr int PD = TD + 3, PH = TH + 3, PW = 20, PHW = PH * PW, NP = PD * PHW;   // patch: origin (d0 - 1, h0 - 1, w0 - 1)
constexpr int kSlabHalf = 4 * 4 * 32;         // entries (16 B) of the hi (or lo) half of a step's slab: <= 4 rows x 4 taps x 32 columns
constexpr int kLdsTab = 256;                  // scale / shift of the 16 input channels
constexpr size_t kLdsFwd = kLdsTab + (size_t)4 * NP * 16 + (size_t)2 * 2 * kSlabHalf * 16;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// the three products of an accumulator as one block of adjacent MFMAs (conv_bf3.hip)
__device__ __forceinline__ void mfma3(f32x4& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
      : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl));
}

struct CtGeom {
  const float* x; long long x_sB; int B, D, H, W;          // forward: input [B][16][D][H][W]; data gradient: dx (output)
  const float* scale; const float* shift; int pre_relu, post_relu;
  const void* wimg;
  const float* bias;
  float* y; long long y_sB, y_sC; int Cout;                // forward: output [B][Cout][2D][2H][2W]; data gradient: dy (input)
  int tilesD, tilesH, tilesW;
  int nhalf;                                               // data gradient: 8-channel halves of Cout (1 or 2)
  int accumulate;
};

// step s of the walk: its slab (hi half, lo half; DH rows of 4 taps x 32 columns each) -> slab buffer `buf`, 1 KiB pieces
template <int DH>
__device__ __forceinline__ void slab_dma(const crn_rsrc& wrs, unsigned img_off, unsigned lds_buf, int wave, int lane) {
  constexpr int kPieces = 4 * DH;                          // 2 DH for hi, 2 DH for lo
#pragma unroll
  for (int j = 0; j < (kPieces + 7) / 8; ++j) {
    const int piece = wave + j * 8;                        // wave-uniform
    if (piece < kPieces) {
      const unsigned off = img_off + (unsigned)piece * 1024u + (unsigned)lane * 16u;
      const unsigned dst = piece < 2 * DH ? lds_buf + (unsigned)piece * 1024u
                                          : lds_buf + (unsigned)(kSlabHalf * 16) + (unsigned)(piece - 2 * DH) * 1024u;
      const unsigned m = __builtin_amdgcn_readfirstlane(dst);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :: "s"(m), "v"(off), "s"(wrs) : "memory");
    }
  }
}

// ------------------------------------------------------------------ forward ------------------------------------
__global__ __launch_bounds__(kT) void convt_par_fwd_kernel(CtGeom g) {
  crn_kernarg_touch(g);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tscale = reinterpret_cast<float*>(smem);
  float* tshift = tscale + 16;
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + kLdsTab);   // [2 chunks][NP]
  bf16x8* Alo = Ahi + 2 * NP;
  bf16x8* Bb = Alo + 2 * NP;                                 // [2 buffers][hi kSlabHalf | lo kSlabHalf]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;

  const crn_rsrc wrs = make_rsrc(reinterpret_cast<const float*>(g.wimg));
  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Bb);
  constexpr unsigned kBufBytes = 2 * kSlabHalf * 16;
  slab_dma<3>(wrs, 0u, lds_b, wave, lane);                   // step 0 = (rd 0, rh 0, chunk 0, plane 0): 3 rows

  if (tid < 16) {
    tscale[tid] = g.scale ? g.scale[tid] : 1.f;
    tshift[tid] = g.scale ? g.shift[tid] : 0.f;
  }
  // ---- patch: unit = (patch row, 16-byte quad of the row), all 16 channels; rows start at w0 - 4 (16-byte aligned) ----
  {
    const crn_rsrc xrs = make_rsrc(g.x + (long long)b * g.x_sB);
    constexpr int kUnits = PD * PH * 6;                      // 462 <= 512
    const int u = tid < kUnits ? tid : 0;
    const int row = u / 6, quad = u - row * 6;
    const int pz = row / PH, py = row - pz * PH;
    const int gd = d0 + pz - 1, gh = h0 + py - 1, gw = w0 - 4 + 4 * quad;
    const bool in = tid < kUnits && (unsigned)gd < (unsigned)g.D && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
    const unsigned sp = ((unsigned)gd * (unsigned)g.H + (unsigned)gh) * (unsigned)g.W + (unsigned)gw;
    const unsigned sC = (unsigned)g.D * (unsigned)g.H * (unsigned)g.W;
    f32x4 pv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) crn_bload4(pv[c], xrs, in ? ((unsigned)c * sC + sp) * 4u : 0x80000000u);
    crn_wait_loads4n(pv);
    __syncthreads();                                         // the channel tables are in LDS
    if (tid < kUnits) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(tscale + ch * 8), s1 = *reinterpret_cast<const f32x4*>(tscale + ch * 8 + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tshift + ch * 8), t1 = *reinterpret_cast<const f32x4*>(tshift + ch * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pl = 4 * quad + j - 3;                   // patch column of element j
          if (pl < 0 || pl >= PW) continue;
          float v[8];
#pragma unroll
          for (int cl = 0; cl < 8; ++cl) {
            float a = pv[ch * 8 + cl][j];
            if (g.scale && in) {                             // zero padding stays zero
              const float sc = cl < 4 ? s0[cl & 3] : s1[cl & 3], sh = cl < 4 ? t0[cl & 3] : t1[cl & 3];
              if (g.pre_relu) a = fmaxf(a, 0.f);
              a = a * sc + sh;
              if (g.post_relu) a = fmaxf(a, 0.f);
            }
            v[cl] = a;
          }
          bf16x8 h, l;
          split8(v, h, l);
          Ahi[ch * NP + row * PW + pl] = h;
          Alo[ch * NP + row * PW + pl] = l;
        }
      }
    }
  }

  // ---- the walk: (rd, rh) pairs x chunks x window planes; sub-tile ms of wave w = row (w & 1) * 4 + ms of plane w >> 1 ----
  const int sd = wave >> 1, sh0 = (wave & 1) * 4;
  const unsigned pa = (unsigned)((sd * PH + sh0) * PW + i16 + kk);
  f32x4 acc[2][4][2];
  unsigned img_off = 0;
  int step = 0;
  const float bsv = (g.bias && i16 < g.Cout) ? g.bias[i16] : 0.f;
  float* const ybase = g.y + (long long)b * g.y_sB + (long long)i16 * g.y_sC;
  const int OH = 2 * g.H, OW = 2 * g.W;

  auto store_pair = [&](f32x4 (&a)[4][2], int rd, int rh) {
    if (i16 < g.Cout) {
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        const int od = 2 * (d0 + sd) + rd, oh = 2 * (h0 + sh0 + ms) + rh, ow = 2 * (w0 + 4 * kk);
        float* dst = ybase + ((long long)od * OH + oh) * OW + ow;
        f32x4 v0 = (f32x4){a[ms][0][0], a[ms][1][0], a[ms][0][1], a[ms][1][1]} + bsv;
        f32x4 v1 = (f32x4){a[ms][0][2], a[ms][1][2], a[ms][0][3], a[ms][1][3]} + bsv;
        *reinterpret_cast<f32x4*>(dst) = v0;
        *reinterpret_cast<f32x4*>(dst + 4) = v1;
      }
    }
  };

  auto walk = [&](auto pp_tag) {
    constexpr int PP = decltype(pp_tag)::value;
    constexpr int RD = PP >> 1, RH = PP & 1, DZ = 3 + RD, DH = 3 + RH;
    constexpr int NDH = PP == 3 ? 3 : (((PP + 1) & 1) ? 4 : 3);          // rows of the step AFTER this pair's last one
    f32x4 (&a)[4][2] = acc[PP & 1];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) { a[ms][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; a[ms][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll 1
      for (int zd = 0; zd < DZ; ++zd) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of this step's slab have landed
        __syncthreads();                                     // ... everybody's; and nobody reads the other buffer any more
        const unsigned nbuf = lds_b + (unsigned)((step + 1) & 1) * kBufBytes;
        const bool last_of_pair = ch == 1 && zd == DZ - 1;
        if constexpr (PP > 0) {
          if (ch == 0 && zd == 0) store_pair(acc[(PP - 1) & 1], (PP - 1) >> 1, (PP - 1) & 1);   // under this step's MFMAs
        }
        img_off += (unsigned)DH * 4096u;
        if (!last_of_pair) slab_dma<DH>(wrs, img_off, nbuf, wave, lane);
        else if (PP < 3) slab_dma<NDH>(wrs, img_off, nbuf, wave, lane);
        const bf16x8* bh0 = Bb + (step & 1) * (2 * kSlabHalf);
        const bf16x8* bl0 = bh0 + kSlabHalf;
        const bf16x8* Ah = Ahi + ch * NP + zd * PHW;
        const bf16x8* Al = Alo + ch * NP + zd * PHW;
#pragma unroll
        for (int zh = 0; zh < DH; ++zh) {
          bf16x8 bh[2], bl[2], ah[4], al[4];
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            bh[ns] = bh0[(zh * 4 + kk) * 32 + ns * 16 + i16];
            bl[ns] = bl0[(zh * 4 + kk) * 32 + ns * 16 + i16];
          }
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) { ah[ms] = Ah[pa + (unsigned)((ms + zh) * PW)]; al[ms] = Al[pa + (unsigned)((ms + zh) * PW)]; }
#pragma unroll
          for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int ns = 0; ns < 2; ++ns) mfma3(a[ms][ns], ah[ms], al[ms], bh[ns], bl[ns]);
        }
        ++step;
      }
    }
  };
  walk(std::integral_constant<int, 0>());
  walk(std::integral_constant<int, 1>());
  walk(std::integral_constant<int, 2>());
  walk(std::integral_constant<int, 3>());
  store_pair(acc[1], 1, 1);
}

// ------------------------------------------------------------------ data gradient ------------------------------
// dx[b, c, q] = sum over (rd, rh, rw), n, window position z of dy[b, n, 2 (q + z - 2) + r] * W[c, n, 2 z - 1 + r]: a stride-1
// correlation over the space-to-depth channels of dy (the origin of the patch is q - 2; parity 0 has no tap at z = 0).  A
// STAGING = (rd, rh, 8-channel half of n): the patch of BOTH rw parities ([rw][position][8 n]) from one 16-byte load per
// (fine row, 2 coarse positions, channel) -- the generic engine loads 12 bytes per channel and position pair and parity --,
// then one step per window plane with the two rw as the two halves of the step's slab; the 16 input channels c are the 16
// MFMA columns and the accumulators live for the whole tile.  The loads of the next staging fly under the steps of this one.
__global__ __launch_bounds__(kT) void convt_par_dgrad_kernel(CtGeom g) {
  crn_kernarg_touch(g);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + kLdsTab);   // [2 rw][NP]
  bf16x8* Alo = Ahi + 2 * NP;
  bf16x8* Bb = Alo + 2 * NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;

  const crn_rsrc wrs = make_rsrc(reinterpret_cast<const float*>(g.wimg));
  const crn_rsrc yrs = make_rsrc(g.y + (long long)b * g.y_sB);
  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Bb);
  constexpr unsigned kBufBytes = 2 * kSlabHalf * 16;
  slab_dma<3>(wrs, 0u, lds_b, wave, lane);                   // step 0 = (rd 0, rh 0, half 0, plane 1): 3 rows

  // ---- patch units: (patch row, 16-byte quad of the fine row = 2 coarse positions x 2 rw), 8 channels; two per thread ----
  constexpr int kUnits = PD * PH * 10;                       // 770
  int urow[2], uquad[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = min(tid + j * kT, kUnits - 1);
    urow[j] = u / 10; uquad[j] = u - urow[j] * 10;
  }
  const int OD = 2 * g.D, OH = 2 * g.H, OW = 2 * g.W;
  f32x4 pv[2][8];
  auto patch_issue = [&](int rd, int rh, int nh) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pz = urow[j] / PH, py = urow[j] - pz * PH;
      const int od = 2 * (d0 - 2 + pz) + rd, oh = 2 * (h0 - 2 + py) + rh, ow = 2 * w0 - 4 + 4 * uquad[j];
      const bool in = tid + j * kT < kUnits && (unsigned)od < (unsigned)OD && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW &&
                      (rd | pz) != 0 && (rh | py) != 0;      // (parity 0 has no tap on the first plane / row of the patch)
      const unsigned sp = ((unsigned)od * (unsigned)OH + (unsigned)oh) * (unsigned)OW + (unsigned)ow;
#pragma unroll
      for (int cl = 0; cl < 8; ++cl) {
        const int n = nh * 8 + cl;
        crn_bload4(pv[j][cl], yrs, (in && n < g.Cout) ? ((unsigned)n * (unsigned)g.y_sC + sp) * 4u : 0x80000000u);
      }
    }
  };
  auto patch_commit = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (tid + j * kT < kUnits) {
        const int base = urow[j] * PW + 2 * uquad[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                        // element e of the quad = (position e >> 1, rw e & 1)
          float v[8];
#pragma unroll
          for (int cl = 0; cl < 8; ++cl) v[cl] = pv[j][cl][e];
          bf16x8 h, l;
          split8(v, h, l);
          Ahi[(e & 1) * NP + base + (e >> 1)] = h;
          Alo[(e & 1) * NP + base + (e >> 1)] = l;
        }
      }
    }
  };

  const int sd = wave >> 1, sh0 = (wave & 1) * 4;
  const unsigned pa = (unsigned)((sd * PH + sh0) * PW + i16 + kk);
  f32x4 acc[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) acc[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned img_off = 0;
  int step = 0;
  patch_issue(0, 0, 0);

  auto walk = [&](auto pp_tag) {
    constexpr int PP = decltype(pp_tag)::value;
    constexpr int RD = PP >> 1, RH = PP & 1, DZ = 3 + RD, DH = 3 + RH, Z0 = 1 - RD, H0 = 1 - RH;
    constexpr int NDH = ((PP + 1) & 1) ? 4 : 3;                         // rows of the steps of the next pair
#pragma unroll 1
    for (int nh = 0; nh < g.nhalf; ++nh) {
      const bool last_staging = PP == 3 && nh == g.nhalf - 1;
      crn_wait_loads4n(pv[0]); crn_wait_loads4n(pv[1]);
      __syncthreads();                                       // nobody reads the patch of the previous staging any more
      patch_commit();
      {                                                      // the loads of the NEXT staging fly under the steps of this one
        const bool same = nh + 1 < g.nhalf;
        const int npp = same ? PP : PP + 1;
        patch_issue(last_staging ? 1 : (npp >> 1), last_staging ? 1 : (npp & 1), last_staging ? 0 : (same ? nh + 1 : 0));
      }
#pragma unroll 1
      for (int zi = 0; zi < DZ; ++zi) {
        const int zd = Z0 + zi;
        // my pieces of this step's slab have landed: they were issued one step ago -- in front of the 16 patch loads when this is
        // the first step of the staging (loads return in order)
        if (zi == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned nbuf = lds_b + (unsigned)((step + 1) & 1) * kBufBytes;
        img_off += (unsigned)DH * 4096u;
        const bool last_of_pair = nh == g.nhalf - 1 && zi == DZ - 1;
        if (!last_of_pair) slab_dma<DH>(wrs, img_off, nbuf, wave, lane);
        else if (PP < 3) slab_dma<NDH>(wrs, img_off, nbuf, wave, lane);
        const bf16x8* bh0 = Bb + (step & 1) * (2 * kSlabHalf);
        const bf16x8* bl0 = bh0 + kSlabHalf;
        const bf16x8* Ah = Ahi + zd * PHW;
        const bf16x8* Al = Alo + zd * PHW;
#pragma unroll
        for (int zi_h = 0; zi_h < DH; ++zi_h) {
          const int zh = H0 + zi_h;
#pragma unroll
          for (int rw = 0; rw < 2; ++rw) {
            const bf16x8 bh = bh0[(zi_h * 4 + kk) * 32 + rw * 16 + i16];
            const bf16x8 bl = bl0[(zi_h * 4 + kk) * 32 + rw * 16 + i16];
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
              ah[ms] = Ah[rw * NP + pa + (unsigned)((ms + zh) * PW)];
              al[ms] = Al[rw * NP + pa + (unsigned)((ms + zh) * PW)];
            }
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) mfma3(acc[ms], ah[ms], al[ms], bh, bl);
          }
        }
        ++step;
      }
    }
  };
  walk(std::integral_constant<int, 0>());
  walk(std::integral_constant<int, 1>());
  walk(std::integral_constant<int, 2>());
  walk(std::integral_constant<int, 3>());

  // epilogue: D row = 4 kk + r = W position of the sub-tile, column i16 = input channel c
  float* dxb = const_cast<float*>(g.x) + (long long)b * g.x_sB + (long long)i16 * ((long long)g.D * g.H * g.W);
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    float* dst = dxb + ((long long)(d0 + sd) * g.H + (h0 + sh0 + ms)) * g.W + w0 + 4 * kk;
    f32x4 v = acc[ms];
    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = v;
  }
}

// ------------------------------------------------------------------ weight image -------------------------------
// dst entry (16 B) table[e][8] = 8 bf16 hi of src[table[e][0..7]] (index < 0: 0), entry table[e][9] their lo parts
__global__ __launch_bounds__(256) void bf3_gather_image_kernel(const float* src, const int* table, int n, char* dst) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int* r = table + (long long)e * 10;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int ix = r[j]; v[j] = ix >= 0 ? src[ix] : 0.f; }
  bf16x8 h, l;
  split8(v, h, l);
  *reinterpret_cast<bf16x8*>(dst + (long long)r[8] * 16) = h;
  *reinterpret_cast<bf16x8*>(dst + (long long)r[9] * 16) = l;
}

}  // namespace

extern "C" int crn_bf3_gather_image(const float* src, const int32_t* table, int n_entries, void* dst, crnStream stream) {
  CRN_ENTRY(stream);
  if (!src || !table || !dst || n_entries < 1) return CRN_EINVAL;
  hipLaunchKernelGGL(bf3_gather_image_kernel, dim3((unsigned)crn_cdiv(n_entries, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, table, n_entries, reinterpret_cast<char*>(dst));
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_convt_s2k7_fwd_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                                      const void* wimg, const float* bias, float* y, int64_t y_sB, int64_t y_sC, int Cout,
                                      crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !wimg || !y || B < 1 || Cout < 1 || Cout > 16) return CRN_EINVAL;
  if (D % TD || H % TH || W % TW || D < TD || (W & 3)) return CRN_EINVAL;
  if ((int64_t)16 * D * H * W >= ((int64_t)1 << 29)) return CRN_EINVAL;          // 32-bit byte offsets inside a sample
  if ((((uintptr_t)x) & 15) || (x_sB & 3) || (((uintptr_t)y) & 15) || (y_sB & 3) || (y_sC & 3)) return CRN_EINVAL;
  CtGeom g{};
  g.x = x; g.x_sB = x_sB; g.B = B; g.D = D; g.H = H; g.W = W;
  if (tr && tr->scale) { g.scale = tr->scale; g.shift = tr->shift; g.pre_relu = tr->pre_relu; g.post_relu = tr->post_relu; }
  g.wimg = wimg; g.bias = bias; g.y = y; g.y_sB = y_sB; g.y_sC = y_sC; g.Cout = Cout;
  g.tilesD = D / TD; g.tilesH = H / TH; g.tilesW = W / TW;
  const int64_t tiles = (int64_t)B * g.tilesD * g.tilesH * g.tilesW;
  if (tiles > 0x7fffffff) return CRN_EINVAL;
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)convt_par_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsFwd) == hipSuccess;
  }();
  if (!attr) return CRN_EINVAL;
  hipLaunchKernelGGL(convt_par_fwd_kernel, dim3((unsigned)tiles), dim3(kT), kLdsFwd, (hipStream_t)stream, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_convt_s2k7_dgrad_bf3(const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, int B, int D, int H, int W,
                                        const void* wimg, float* dx, int64_t dx_sB, int accumulate, crnStream stream) {
  CRN_ENTRY(stream);
  if (!dy || !wimg || !dx || B < 1 || Cout < 1 || Cout > 16) return CRN_EINVAL;
  if (D % TD || H % TH || W % TW || D < TD) return CRN_EINVAL;
  if ((int64_t)Cout * dy_sC >= ((int64_t)1 << 29) || dy_sC < (int64_t)8 * D * H * W) return CRN_EINVAL;   // 32-bit byte offsets
  if ((((uintptr_t)dy) & 15) || (dy_sB & 3) || (dy_sC & 3) || (((uintptr_t)dx) & 15) || (dx_sB & 3)) return CRN_EINVAL;
  CtGeom g{};
  g.x = dx; g.x_sB = dx_sB; g.B = B; g.D = D; g.H = H; g.W = W;
  g.wimg = wimg; g.y = const_cast<float*>(dy); g.y_sB = dy_sB; g.y_sC = dy_sC; g.Cout = Cout;
  g.nhalf = (Cout + 7) / 8; g.accumulate = accumulate ? 1 : 0;
  g.tilesD = D / TD; g.tilesH = H / TH; g.tilesW = W / TW;
  const int64_t tiles = (int64_t)B * g.tilesD * g.tilesH * g.tilesW;
  if (tiles > 0x7fffffff) return CRN_EINVAL;
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)convt_par_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsFwd) == hipSuccess;
  }();
  if (!attr) return CRN_EINVAL;
  hipLaunchKernelGGL(convt_par_dgrad_kernel, dim3((unsigned)tiles), dim3(kT), kLdsFwd, (hipStream_t)stream, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
