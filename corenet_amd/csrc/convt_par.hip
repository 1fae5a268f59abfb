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
  int xcd;                                                 // resident-weights kernels: XCD-aware tile ranges (CRN_CT_RES_XCD=0: off)
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

// ------------------------------------------------------------------ weight gradient ----------------------------
// dW[c, n, k] = sum over (b, q) of T(x)[b, c, q + z - 1] * dy[b, n, 2 q + r],  k = 5 - 2 z + r: per output parity r a GEMM
// D_z[c][n] = X_z^T . dY_r with K = positions, like conv_bf3_wgrad_kernel (conv_bf3.hip) -- 32 positions per MFMA, both
// operands read out of position-major LDS images with the transposing ds_read_b64_tr_b16 -- but cut the way the layer is:
// a workgroup owns ONE (rd, rh) pair (a range of blockIdx.x: the pairs get workgroups in proportion to their cost), both rw (2 x 16 columns), ALL 16 input channels (the MFMA's 16 rows = 2
// chunks x 8 channels of one tap instead of 2 taps x 8 channels) and a slice of the 2 x 8 x 16 position tiles.
// So dy is read once in total (each pair's workgroups load only their own fine rows, 16-byte loads that serve both rw), the
// taps are exactly the (3 + rd)(3 + rh) 4 of the pair's box, dealt round-robin to the 8 waves, and nothing is multiplied for
// columns of another parity (the generic kernel's 16-column blocks straddle parities at 14 classes: 956 us at B = 4).
// Partial sums go to the layer's packed gradient [c][64 window taps][Npad] (parity-major columns, conv_geometry.convt_fwd)
// with fire-and-forget atomics, like the generic kernel: the un-pack is unchanged.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr int kYSwz = 1;                                      // 1: bank swizzle of the dy image (0: the plain layout, for A/B runs of the build)
constexpr int WTD = 2, WPD = WTD + 3, WNP = WPD * PHW;       // weight-gradient tile: 2 x 8 x 16 positions = 8 K blocks of 32
constexpr int kWgPos = WTD * TH * TW;                        // 256
constexpr size_t kLdsWg = kLdsTab + (size_t)4 * WNP * 16 + (size_t)2 * kWgPos * 32 * 2;

struct CtWgGeom {
  const float* x; long long x_sB; int B, D, H, W;
  const float* scale; const float* shift; int pre_relu, post_relu;
  const float* dy; long long dy_sB, dy_sC; int Cout;
  const float* ximg; long long ximg_sB;                      // operand image of T(x) (convt_ximage_kernel): floats per sample, or nullptr
  float* dw; int Npad;
  int tilesD, tilesH, tilesW, ntiles;
  int wg_end[4], tiles_per_wg[4];                            // workgroups [wg_end[p - 1], wg_end[p]) walk pair p, tiles_per_wg[p] tiles each
};

template <int TPW, bool XIMG>
__device__ __forceinline__ void convt_par_wgrad_body(const CtWgGeom& g, int pp, int slot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tscale = reinterpret_cast<float*>(smem);
  float* tshift = tscale + 16;
  char* Xhi = smem + kLdsTab;                                // [2 chunks][WNP] entries of 8 bf16
  char* Xlo = Xhi + (size_t)2 * WNP * 16;
  char* Yhi = Xlo + (size_t)2 * WNP * 16;                    // [256 positions][32 columns] bf16
  char* Ylo = Yhi + (size_t)kWgPos * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;
  const int rd = pp >> 1, rh = pp & 1, dz = 3 + rd, dh = 3 + rh;
  const int ntaps = dz * dh * 4;
  const int tbeg = min(slot * g.tiles_per_wg[pp], g.ntiles), tend = min(tbeg + g.tiles_per_wg[pp], g.ntiles);

  if (tid < 16) {
    tscale[tid] = g.scale ? g.scale[tid] : 1.f;
    tshift[tid] = g.scale ? g.shift[tid] : 0.f;
  }
  // transposing reads: lane = row j of the [4 K][16 M] block of its 16-lane group, column quad q (conv_bf3_wgrad_kernel)
  const int j = i16 >> 2, q = i16 & 3;
  const int k1 = kk * 8 + j;                                 // K index inside the 32-position block (second read: + 4)
  // A (input patch): column quad q = (chunk q >> 1, channel half q & 1) of the 16 input channels
  const int abase = (((k1 >> 4) * PW + (k1 & 15)) << 4) + ((q & 1) << 3) + (q >> 1) * (WNP * 16);
  // TPW = taps per wave = ceil(ntaps / 8): 36 / 48 / 48 / 64 taps -> 5 / 6 / 6 / 8 (compile time: the tap loop is straight-line
  // code and software-pipelined).  With 36 taps four waves have no fifth one: they multiply the clamped last tap (zw = 3: one
  // triple) once more and drop the result
  // Round ti deals taps 8 ti .. 8 ti + 7 to the waves ROTATED by ti (wave w takes 8 ti + ((w + ti) & 7)): the window column zw = tap & 3
  // then walks through a wave's taps instead of being the wave's own, and the taps with zw = 3 -- which have no weight for rw = 0
  // (k = 5 - 2 z + r < 0) and whose rw = 0 triple is therefore skipped, an eighth of all MFMAs -- are spread evenly over the waves
  int toffL[TPW];
  bool skip0[TPW];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int tp = min(8 * ti + ((wave + ti) & 7), ntaps - 1);
    const int zw = tp & 3, zr = tp >> 2, zh = zr % dh, zd = zr / dh;
    toffL[ti] = ((zd * PH + zh) * PW + zw) << 4;
    skip0[ti] = zw == 3;
  }
  const int ybase = k1 * 64 + (q << 3);
  const int yswz = (kk & kYSwz) * 32;
  constexpr int xlo = 2 * WNP * 16, ylo = kWgPos * 64;

  f32x4 acc[TPW][2];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) { acc[ti][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[ti][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // staging units.  x: (patch row, 16-byte quad) x 16 channels, threads 0 .. 329; dy: (position pair, 8-channel half of n) of
  // the pair's fine rows -- one 16-byte load = 2 positions x 2 rw --, threads 256 .. 511
  constexpr int kXUnits = WPD * PH * 6;                      // 330
  const int xu = tid < kXUnits ? tid : 0;
  const int xrow = xu / 6, xquad = xu - xrow * 6;
  const int xpz = xrow / PH, xpy = xrow - xpz * PH;
  const int du = (tid - 256) & 255, dpq = du & 127, doct = du >> 7;
  const int dp = dpq * 2, dwp = dp & 15, dhp = (dp >> 4) & 7, ddp = dp >> 7;
  f32x4 pv[16], dv[8];
  bool xin = false;
  int b = 0, d0 = 0, h0 = 0, w0 = 0;
  const int OH = 2 * g.H, OW = 2 * g.W;
  auto tile_origin = [&](int tl) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    b = tile; d0 = tdi * WTD; h0 = thi * TH; w0 = twi * TW;
  };
  auto stage_issue = [&]() {
    const crn_rsrc xrs = make_rsrc(g.x + (long long)b * g.x_sB);
    const crn_rsrc drs = make_rsrc(g.dy + (long long)b * g.dy_sB);
    const int gd = d0 + xpz - 1, gh = h0 + xpy - 1, gw = w0 - 4 + 4 * xquad;
    xin = tid < kXUnits && (unsigned)gd < (unsigned)g.D && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
    const unsigned sp = ((unsigned)gd * (unsigned)g.H + (unsigned)gh) * (unsigned)g.W + (unsigned)gw;
    const unsigned sC = (unsigned)g.D * (unsigned)g.H * (unsigned)g.W;
    if constexpr (XIMG) {
      // the input arrives transformed and split (convt_ximage_kernel): [region = chunk * 2 + (hi, lo)][position] entries of 8 bf16;
      // register r * 4 + e = region r, position e of the unit's four -- the commit is a plain copy
      const crn_rsrc irs = make_rsrc(g.ximg + (long long)b * g.ximg_sB);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pl = 4 * xquad + e - 3;
          crn_bload4(pv[r * 4 + e], irs, (xin && pl >= 0 && pl < PW) ? ((unsigned)r * sC + sp + (unsigned)e) * 16u : 0x80000000u);
        }
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) crn_bload4(pv[c], xrs, xin ? ((unsigned)c * sC + sp) * 4u : 0x80000000u);
    }
    const unsigned od = 2 * (d0 + ddp) + rd, oh = 2 * (h0 + dhp) + rh, ow = 2 * (w0 + dwp);
    const unsigned dsp = (od * (unsigned)OH + oh) * (unsigned)OW + ow;
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) {
      const int n = doct * 8 + cl;
      crn_bload4(dv[cl], drs, (tid >= 256 && n < g.Cout) ? ((unsigned)n * (unsigned)g.dy_sC + dsp) * 4u : 0x80000000u);
    }
  };
  auto stage_commit = [&]() {
    if constexpr (XIMG) {
      if (tid < kXUnits) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pl = 4 * xquad + e - 3;
          if (pl < 0 || pl >= PW) continue;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            *reinterpret_cast<f32x4*>(Xhi + ((size_t)ch * WNP + xrow * PW + pl) * 16) = pv[(ch * 2 + 0) * 4 + e];
            *reinterpret_cast<f32x4*>(Xlo + ((size_t)ch * WNP + xrow * PW + pl) * 16) = pv[(ch * 2 + 1) * 4 + e];
          }
        }
      }
    } else
    if (tid < kXUnits) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(tscale + ch * 8), s1 = *reinterpret_cast<const f32x4*>(tscale + ch * 8 + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tshift + ch * 8), t1 = *reinterpret_cast<const f32x4*>(tshift + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pl = 4 * xquad + e - 3;
          if (pl < 0 || pl >= PW) continue;
          float v[8];
#pragma unroll
          for (int cl = 0; cl < 8; ++cl) {
            float a = pv[ch * 8 + cl][e];
            if (g.scale && xin) {
              const float sc = cl < 4 ? s0[cl & 3] : s1[cl & 3], sh = cl < 4 ? t0[cl & 3] : t1[cl & 3];
              if (g.pre_relu) a = fmaxf(a, 0.f);
              a = a * sc + sh;
              if (g.post_relu) a = fmaxf(a, 0.f);
            }
            v[cl] = a;
          }
          bf16x8 h, l;
          split8(v, h, l);
          *reinterpret_cast<bf16x8*>(Xhi + ((size_t)ch * WNP + xrow * PW + pl) * 16) = h;
          *reinterpret_cast<bf16x8*>(Xlo + ((size_t)ch * WNP + xrow * PW + pl) * 16) = l;
        }
      }
    }
    if (tid >= 256) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {                            // element e = (position e >> 1 of the pair, rw e & 1)
        float v[8];
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) v[cl] = dv[cl][e];
        bf16x8 h, l;
        split8(v, h, l);
        // (the two 32-byte halves of a position's row -- rw 0 / rw 1 -- swap places on positions with bit 3 set: the transposing
        // reads of lanes kk and kk + 1, eight positions = 512 bytes apart, then fall into different banks; without it every
        // B-fragment read was a 2-way bank conflict, SQ_LDS_BANK_CONFLICT 46 % of the kernel's LDS cycles)
        const int pos = dp + (e >> 1);
        const size_t o = (size_t)pos * 64 + (size_t)((((e & 1) ^ ((pos >> 3) & kYSwz)) * 16 + doct * 8) * 2);
        *reinterpret_cast<bf16x8*>(Yhi + o) = h;
        *reinterpret_cast<bf16x8*>(Ylo + o) = l;
      }
    }
  };
  auto trd = [&](const char* p) -> bf16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p)); };
  auto cat = [](bf16x4 a, bf16x4 c) -> bf16x8 { return __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7); };

  if (tbeg < tend) { tile_origin(tbeg); stage_issue(); }
  for (int tl = tbeg; tl < tend; ++tl) {
    crn_wait_loads4n(pv);
    crn_wait_loads4n(dv);
    __syncthreads();
    stage_commit();
    __syncthreads();
    if (tl + 1 < tend) { tile_origin(tl + 1); stage_issue(); }
    // Software-pipelined over (K block kb, tap ti): the transposing reads of the NEXT unit's A fragments (at the last tap of a K
    // block: the next block's B fragments too) are issued before the MFMAs of the current one -- two waves per SIMD cannot hide an LDS
    // round trip per MFMA pair otherwise (conv_bf3_wgrad_kernel, profiles/r03_wgrad_pipe_ab.txt)
    auto kbx_of = [&](int kb) { return (((kb >> 2) * PH + (kb & 3) * 2) * PW) << 4; };      // K block kb = (plane, H row pair)
    bf16x8 bh[2][2], bl[2][2], ah[2], al[2];
    auto ldB = [&](int kb, int qb) {
      const char* yb = Yhi + ybase + kb * (32 * 64);
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        const int co = (ns * 32) ^ yswz;                     // (this lane's positions 8 kk + j [+ 4] have bit 3 = kk & 1)
        bh[qb][ns] = cat(trd(yb + co), trd(yb + co + 4 * 64));
        bl[qb][ns] = cat(trd(yb + ylo + co), trd(yb + ylo + co + 4 * 64));
      }
    };
    auto ldA = [&](int kbx, int ti, int qa) {
      const char* xa = Xhi + abase + toffL[ti] + kbx;
      ah[qa] = cat(trd(xa), trd(xa + 64));
      al[qa] = cat(trd(xa + xlo), trd(xa + xlo + 64));
    };
    auto block = [&](int kb, auto QB_) {
      constexpr int qb = decltype(QB_)::value;
      const int kbx = kbx_of(kb);
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int qa = (TPW & 1) ? ((qb * TPW + ti) & 1) : (ti & 1);       // A buffers alternate across the whole stream
        if (ti + 1 < TPW) ldA(kbx, ti + 1, qa ^ 1);
        else { const int kn = min(kb + 1, 7); ldB(kn, qb ^ 1); ldA(kbx_of(kn), 0, qa ^ 1); }
        __builtin_amdgcn_sched_barrier(0);
        if (!skip0[ti]) mfma3(acc[ti][0], ah[qa], al[qa], bh[qb][0], bl[qb][0]);
        mfma3(acc[ti][1], ah[qa], al[qa], bh[qb][1], bl[qb][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    ldB(0, 0);
    ldA(kbx_of(0), 0, 0);
#pragma unroll 1
    for (int kb = 0; kb < 8; kb += 2) {
      block(kb, std::integral_constant<int, 0>());
      block(kb + 1, std::integral_constant<int, 1>());
    }
  }

  // D row m = 4 kk + r = input channel c, column i16 = n of block ns = rw; packed gradient [c][(zd * 4 + zh) * 4 + zw][Npad]
  if (i16 < g.Cout) {
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tp = 8 * ti + ((wave + ti) & 7);
      if (tp >= ntaps) continue;                               // (the clamped duplicates of the last round at 36 taps)
      const int zw = tp & 3, zr = tp >> 2, zh = zr % dh, zd = zr / dh;
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        if (zw == 3 && ns == 0) continue;                      // (k = 5 - 2 z + r < 0: no such tap for rw = 0)
        const int col = ((rd * 2 + rh) * 2 + ns) * g.Cout + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = kk * 4 + r;
          atomicAdd(g.dw + ((long long)c * 64 + (zd * 4 + zh) * 4 + zw) * g.Npad + col, acc[ti][ns][r]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(kT) void convt_par_wgrad_kernel(CtWgGeom g) {
  crn_kernarg_touch(g);
  // the (rd, rh) pair of this workgroup: 36 / 48 / 48 / 64 taps.  The pairs get workgroups in proportion to what a tile costs them
  // (host side): with the same number of tiles per workgroup for every pair the launch lasted as long as pair 3's 64 taps
  const int wg = blockIdx.x;
  const int pp = wg < g.wg_end[0] ? 0 : wg < g.wg_end[1] ? 1 : wg < g.wg_end[2] ? 2 : 3;
  const int slot = wg - (pp ? g.wg_end[pp - 1] : 0);
  if (g.ximg) {
    switch (pp) {
      case 0: convt_par_wgrad_body<5, true>(g, 0, slot); break;
      case 3: convt_par_wgrad_body<8, true>(g, 3, slot); break;
      default: convt_par_wgrad_body<6, true>(g, pp, slot); break;
    }
    return;
  }
  switch (pp) {
    case 0: convt_par_wgrad_body<5, false>(g, 0, slot); break;
    case 3: convt_par_wgrad_body<8, false>(g, 3, slot); break;
    default: convt_par_wgrad_body<6, false>(g, pp, slot); break;
  }
}

// ------------------------------------------------------------------ operand image of the layer's input ----------
// The weight-gradient workgroups of the four (rd, rh) pairs all stage the same patches of T(x) = BatchRenorm + ReLU of the layer's
// input, each with a 4.3x halo: transformed and split into bf16 hi / lo in every one of them, the conversion was ~400 VALU
// instructions per staging unit and tile -- about as long as a tile's MFMAs.  One pass over x does it once: img[b][region = chunk * 2
// + (hi, lo)][position] = entry of 8 bf16 (the 8 channels of the chunk), the same transform expression for expression and the same
// split as the fused staging (bit-identical operands).  67 MB read, 67 MB written at B = 4.
__global__ __launch_bounds__(256) void convt_ximage_kernel(const float* x, long long x_sB, int S4, const float* scale, const float* shift,
                                                           int pre_relu, int post_relu, float* img, long long img_sB) {
  const int q = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, b = blockIdx.z;
  if (q >= S4) return;
  const long long S = (long long)S4 * 4;
  const float* xp = x + (long long)b * x_sB + (long long)(ch * 8) * S + (long long)q * 4;
  f32x4 v[8];
#pragma unroll
  for (int cl = 0; cl < 8; ++cl) v[cl] = *reinterpret_cast<const f32x4*>(xp + (long long)cl * S);
  float sc[8], sh[8];
#pragma unroll
  for (int cl = 0; cl < 8; ++cl) { sc[cl] = scale ? scale[ch * 8 + cl] : 1.f; sh[cl] = scale ? shift[ch * 8 + cl] : 0.f; }
  float* hi = img + (long long)b * img_sB + ((long long)(ch * 2) * S + (long long)q * 4) * 4;      // (entries of 4 floats)
  float* lo = hi + S * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a[8];
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) {
      float t = v[cl][e];
      if (scale) {
        if (pre_relu) t = fmaxf(t, 0.f);
        t = t * sc[cl] + sh[cl];
        if (post_relu) t = fmaxf(t, 0.f);
      }
      a[cl] = t;
    }
    bf16x8 h, l;
    split8(a, h, l);
    *reinterpret_cast<bf16x8*>(hi + e * 4) = h;
    *reinterpret_cast<bf16x8*>(lo + e * 4) = l;
  }
}

// ------------------------------------------------------------------ two classes: resident weights ----------------------
// With 2 classes (h7) the 8 parities x 2 classes are ONE 16-column block and the whole layer's weights -- 2 chunks x 4^3 taps
// x 16 columns, 64 KB as bf16 hi / lo -- fit in LDS next to the tile's patch.  The generic engine re-stages a 4 KB weight slab
// and meets at two barriers for every 16 MFMA triples of a wave there (forward 151 us, data gradient 159 us + the fused sums;
// 0.18 / 0.11 of the roof).  Here a PERSISTENT workgroup (one per CU) loads the weights once by LDS-DMA and walks its tiles:
// per tile two barriers around the patch commit, the next tile's loads in flight under this tile's 112-128 triples per wave.
//   forward       : chunk = 8 of the 16 input channels, columns (rd, rh, rw, n); the pixel-shuffle store pairs the rw = 0 / 1
//                   lanes with one DPP exchange so that every lane writes 16 bytes.
//   data gradient : chunk = rd, the 8 channels of an entry are (rh, rw, n) of dy -- one 16-byte load = 2 positions x 2 rw --,
//                   columns = the 16 input channels; rd = 0 has no tap on window plane 0.
constexpr int RPW = 19, RPHW = PH * RPW, RNP = PD * RPHW;    // (19 columns: patch + weights stay under 160 KiB)
constexpr int kResW = 2 * 4 * 4 * 4 * 16;                    // entries of the hi (or lo) half of the resident weights
constexpr size_t kLdsRes = kLdsTab + (size_t)4 * RNP * 16 + (size_t)2 * kResW * 16;

template <bool DGRAD>
__global__ __launch_bounds__(kT) void convt_res_kernel(CtGeom g, int ntiles, int tiles_per_wg) {
  crn_kernarg_touch(g);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tscale = reinterpret_cast<float*>(smem);
  float* tshift = tscale + 16;
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + kLdsTab);   // [2 chunks][RNP]
  bf16x8* Alo = Ahi + 2 * RNP;
  bf16x8* Bhi = Alo + 2 * RNP;                               // [chunk][zd][zh][tap][16 columns]
  bf16x8* Blo = Bhi + kResW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;
  {
    const crn_rsrc wrs = make_rsrc(reinterpret_cast<const float*>(g.wimg));
    const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Bhi);
#pragma unroll
    for (int jp = 0; jp < 8; ++jp) {                         // 64 pieces of 1 KiB: the image IS the LDS layout (hi, then lo)
      const int piece = wave * 8 + jp;
      const unsigned m = __builtin_amdgcn_readfirstlane(lds_w + (unsigned)piece * 1024u);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :: "s"(m), "v"((unsigned)piece * 1024u + (unsigned)lane * 16u), "s"(wrs) : "memory");
    }
  }
  if (!DGRAD && tid < 16) {
    tscale[tid] = g.scale ? g.scale[tid] : 1.f;
    tshift[tid] = g.scale ? g.shift[tid] : 0.f;
  }
  // (XCD-aware: workgroup w runs on XCD w % 8; each XCD walks a contiguous eighth of the tiles so that the patch halos of the
  // workgroups that run side by side hit in ITS L2 -- 520 / 406 MB of fabric traffic per launch for 134 MB of tensors without it)
  const int slot = g.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tbeg = min(slot * tiles_per_wg, ntiles), tend = min(tbeg + tiles_per_wg, ntiles);
  int b = 0, d0 = 0, h0 = 0, w0 = 0;
  auto tile_origin = [&](int tl) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    b = tile; d0 = tdi * TD; h0 = thi * TH; w0 = twi * TW;
  };
  const int OD = 2 * g.D, OH = 2 * g.H, OW = 2 * g.W;
  // ---- staging ----
  // forward: unit = (patch row, 16-byte quad from w0 - 4), 16 channels; 462 units.  data gradient: unit = (chunk rd, patch row,
  // position pair), loads (rh, n); rows of chunk 1 (rd = 1: 7 planes) first, then chunk 0 (planes 1 .. 6): 1430 units, 3 per thread
  constexpr int kNL = DGRAD ? 12 : 16;
  f32x4 pv[kNL];
  bool xin = false;
  constexpr int kFUnits = PD * PH * 6, kDRows1 = PD * PH, kDUnits = (PD + PD - 1) * PH * 10;
  auto stage_issue = [&]() {
    if constexpr (!DGRAD) {
      const crn_rsrc xrs = make_rsrc(g.x + (long long)b * g.x_sB);
      const int u = tid < kFUnits ? tid : 0;
      const int row = u / 6, quad = u - row * 6;
      const int pz = row / PH, py = row - pz * PH;
      const int gd = d0 + pz - 1, gh = h0 + py - 1, gw = w0 - 4 + 4 * quad;
      xin = tid < kFUnits && (unsigned)gd < (unsigned)g.D && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
      const unsigned sp = ((unsigned)gd * (unsigned)g.H + (unsigned)gh) * (unsigned)g.W + (unsigned)gw;
      const unsigned sC = (unsigned)g.D * (unsigned)g.H * (unsigned)g.W;
#pragma unroll
      for (int c = 0; c < 16; ++c) crn_bload4(pv[c], xrs, xin ? ((unsigned)c * sC + sp) * 4u : 0x80000000u);
    } else {
      const crn_rsrc yrs = make_rsrc(g.y + (long long)b * g.y_sB);
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const int u = min(tid + jx * kT, kDUnits - 1);
        const int r = u / 10, pair = u - r * 10;
        const int rd = r < kDRows1 ? 1 : 0, rr = rd ? r : r - kDRows1 + PH;           // patch row (plane * PH + py)
        const int pz = rr / PH, py = rr - pz * PH;
        const int od = 2 * (d0 - 2 + pz) + rd, ow = 2 * w0 - 4 + 4 * pair;
        const bool inz = tid + jx * kT < kDUnits && (unsigned)od < (unsigned)OD && (unsigned)ow < (unsigned)OW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                          // e = (rh, n)
          const int rh = e >> 1, n = e & 1;
          const int oh = 2 * (h0 - 2 + py) + rh;
          const bool in = inz && (unsigned)oh < (unsigned)OH && n < g.Cout;
          crn_bload4(pv[jx * 4 + e], yrs, in ? ((unsigned)n * (unsigned)g.y_sC + ((unsigned)od * (unsigned)OH + (unsigned)oh) * (unsigned)OW + (unsigned)ow) * 4u
                                             : 0x80000000u);
        }
      }
    }
  };
  auto stage_commit = [&]() {
    if constexpr (!DGRAD) {
      if (tid < kFUnits) {
        const int row = tid / 6, quad = tid - row * 6;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(tscale + ch * 8), s1 = *reinterpret_cast<const f32x4*>(tscale + ch * 8 + 4);
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tshift + ch * 8), t1 = *reinterpret_cast<const f32x4*>(tshift + ch * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int pl = 4 * quad + e - 3;
            if (pl < 0 || pl >= RPW) continue;
            float v[8];
#pragma unroll
            for (int cl = 0; cl < 8; ++cl) {
              float a = pv[ch * 8 + cl][e];
              if (g.scale && xin) {
                const float sc = cl < 4 ? s0[cl & 3] : s1[cl & 3], sh = cl < 4 ? t0[cl & 3] : t1[cl & 3];
                if (g.pre_relu) a = fmaxf(a, 0.f);
                a = a * sc + sh;
                if (g.post_relu) a = fmaxf(a, 0.f);
              }
              v[cl] = a;
            }
            bf16x8 h, l;
            split8(v, h, l);
            Ahi[ch * RNP + row * RPW + pl] = h;
            Alo[ch * RNP + row * RPW + pl] = l;
          }
        }
      }
    } else {
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const int u = tid + jx * kT;
        if (u < kDUnits) {
          const int r = u / 10, pair = u - r * 10;
          const int rd = r < kDRows1 ? 1 : 0, rr = rd ? r : r - kDRows1 + PH;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            if (2 * pair + px >= RPW) continue;
            float v[8];
#pragma unroll
            for (int jc = 0; jc < 8; ++jc) {                  // channel jc of the entry = (rh, rw, n)
              const int rh = jc >> 2, rw = (jc >> 1) & 1, n = jc & 1;
              v[jc] = pv[jx * 4 + rh * 2 + n][2 * px + rw];
            }
            bf16x8 h, l;
            split8(v, h, l);
            Ahi[rd * RNP + rr * RPW + 2 * pair + px] = h;
            Alo[rd * RNP + rr * RPW + 2 * pair + px] = l;
          }
        }
      }
    }
  };

  const int sd = wave >> 1, sh0 = (wave & 1) * 4;
  const unsigned pa = (unsigned)((sd * PH + sh0) * RPW + i16 + kk);
  const float bsv = (!DGRAD && g.bias) ? g.bias[i16 & 1] : 0.f;      // (2 classes: column = parity * 2 + n)
  if (tbeg < tend) { tile_origin(tbeg); stage_issue(); }
  for (int tl = tbeg; tl < tend; ++tl) {
#pragma unroll
    for (int i = 0; i < kNL; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[i]));
    __syncthreads();                                         // nobody reads the previous tile's patch any more (first trip: the tables and weights are in LDS)
    stage_commit();
    __syncthreads();
    const int cb = b, cd0 = d0, ch0 = h0, cw0 = w0;
    if (tl + 1 < tend) { tile_origin(tl + 1); stage_issue(); }
    f32x4 acc[4];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) acc[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll 1
      for (int zd = DGRAD ? 1 - c : 0; zd < 4; ++zd) {
        const bf16x8* Ah = Ahi + c * RNP + zd * RPHW;
        const bf16x8* Al = Alo + c * RNP + zd * RPHW;
        const bf16x8* bh0 = Bhi + (c * 4 + zd) * 256;
        const bf16x8* bl0 = Blo + (c * 4 + zd) * 256;
#pragma unroll
        for (int zh = 0; zh < 4; ++zh) {
          const bf16x8 bh = bh0[(zh * 4 + kk) * 16 + i16], bl = bl0[(zh * 4 + kk) * 16 + i16];
          bf16x8 ah[4], al[4];
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) { ah[ms] = Ah[pa + (unsigned)((ms + zh) * RPW)]; al[ms] = Al[pa + (unsigned)((ms + zh) * RPW)]; }
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) mfma3(acc[ms], ah[ms], al[ms], bh, bl);
        }
      }
    }
    if constexpr (DGRAD) {
      float* dxb = const_cast<float*>(g.x) + (long long)cb * g.x_sB + (long long)i16 * ((long long)g.D * g.H * g.W);
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        float* dst = dxb + ((long long)(cd0 + sd) * g.H + (ch0 + sh0 + ms)) * g.W + cw0 + 4 * kk;
        f32x4 v = acc[ms];
        if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    } else {
      // column i16 = ((rd * 2 + rh) * 2 + rw) * 2 + n.  The lane pair (rw 0, rw 1) = (lane, lane ^ 2) holds the two interleaved
      // halves of the same 8 output floats: rw 0 keeps positions 0, 1 and takes the partner's, rw 1 keeps 2, 3 -- one 16-byte store each
      const int rw = (i16 >> 1) & 1, n = i16 & 1, rh = (i16 >> 2) & 1, rd = i16 >> 3;
      if (n < g.Cout) {
        float* yb = g.y + (long long)cb * g.y_sB + (long long)n * g.y_sC;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const f32x4 a = acc[ms] + bsv;
          const float s0 = rw ? a[0] : a[2], s1 = rw ? a[1] : a[3];           // what the partner needs
          const float p0 = __shfl_xor(s0, 2), p1 = __shfl_xor(s1, 2);
          const f32x4 v = rw ? (f32x4){p0, a[2], p1, a[3]} : (f32x4){a[0], p0, a[1], p1};
          const int od = 2 * (cd0 + sd) + rd, oh = 2 * (ch0 + sh0 + ms) + rh, ow = 2 * (cw0 + 4 * kk) + 4 * rw;
          *reinterpret_cast<f32x4*>(yb + ((long long)od * OH + oh) * OW + ow) = v;
        }
      }
    }
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

extern "C" size_t crn_convt_s2k7_ximage_bytes(int B, int D, int H, int W) {
  return (size_t)B * 4 * D * H * W * 16;
}

extern "C" int crn_convt_s2k7_ximage(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                                     void* img, size_t img_bytes, crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !img || B < 1 || D < 1 || H < 1 || W < 1 || (W & 3)) return CRN_EINVAL;
  if (img_bytes < crn_convt_s2k7_ximage_bytes(B, D, H, W)) return CRN_ENOMEM;
  if ((int64_t)16 * D * H * W >= ((int64_t)1 << 29)) return CRN_EINVAL;          // 32-bit byte offsets inside a sample of the image
  if ((((uintptr_t)x) & 15) || (x_sB & 3) || (((uintptr_t)img) & 15)) return CRN_EINVAL;
  const int S4 = D * H * W / 4;
  const bool has = tr && tr->scale;
  hipLaunchKernelGGL(convt_ximage_kernel, dim3((unsigned)crn_cdiv(S4, 256), 2, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     x, (long long)x_sB, S4, has ? tr->scale : nullptr, has ? tr->shift : nullptr, has ? tr->pre_relu : 0,
                     has ? tr->post_relu : 0, reinterpret_cast<float*>(img), (long long)16 * D * H * W);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_convt_s2k7_wgrad_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                                        const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, float* dw, int Npad,
                                        int zero_first, const void* ximg, crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !dy || !dw || B < 1 || Cout < 1 || Cout > 16 || Npad < 8 * Cout) return CRN_EINVAL;
  if (ximg && (((uintptr_t)ximg) & 15)) return CRN_EINVAL;
  if (D % WTD || H % TH || W % TW) return CRN_EINVAL;
  if (crn_deterministic()) return CRN_EINVAL;                 // (split sums with atomics: the caller takes the generic engine)
  if ((int64_t)16 * D * H * W >= ((int64_t)1 << 29) || (int64_t)Cout * dy_sC >= ((int64_t)1 << 29)) return CRN_EINVAL;
  if ((((uintptr_t)x) & 15) || (x_sB & 3) || (((uintptr_t)dy) & 15) || (dy_sB & 3) || (dy_sC & 3)) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)16 * 64 * Npad * sizeof(float), st));
  CtWgGeom g{};
  g.x = x; g.x_sB = x_sB; g.B = B; g.D = D; g.H = H; g.W = W;
  if (tr && tr->scale) { g.scale = tr->scale; g.shift = tr->shift; g.pre_relu = tr->pre_relu; g.post_relu = tr->post_relu; }
  g.dy = dy; g.dy_sB = dy_sB; g.dy_sC = dy_sC; g.Cout = Cout; g.dw = dw; g.Npad = Npad;
  g.ximg = reinterpret_cast<const float*>(ximg); g.ximg_sB = (long long)16 * D * H * W;      // (floats per sample: 4 regions x 4 floats per entry)
  g.tilesD = D / WTD; g.tilesH = H / TH; g.tilesW = W / TW;
  const int64_t ntiles = (int64_t)B * g.tilesD * g.tilesH * g.tilesW;
  if (ntiles > 0x7fffffff) return CRN_EINVAL;
  g.ntiles = (int)ntiles;
  // One round of 256 workgroups (one per CU), dealt to the four (rd, rh) pairs in proportion to the cost of a tile: its multiplies
  // (taps per wave x 8 waves: 40 / 48 / 48 / 64) plus the staging of the tile, which is the same for every pair (kStage, in taps:
  // CRN_CT_WG_STAGE; 0 = by multiplies alone, CRN_CT_WG_EVEN=1 = the same number of tiles for every pair)
  static const int kBlocks = getenv("CRN_CT_WG_BLOCKS") ? std::max(4, atoi(getenv("CRN_CT_WG_BLOCKS"))) : 256;
  static const int kStage = getenv("CRN_CT_WG_STAGE") ? std::max(0, atoi(getenv("CRN_CT_WG_STAGE"))) : 24;
  static const bool kEven = getenv("CRN_CT_WG_EVEN") != nullptr && atoi(getenv("CRN_CT_WG_EVEN")) != 0;
  const int taps[4] = {40, 48, 48, 64};
  int cost[4], total = 0;
  for (int p = 0; p < 4; ++p) { cost[p] = kEven ? 1 : taps[p] + kStage; total += cost[p]; }
  const int blocks = (int)std::min<int64_t>(kBlocks, 4 * ntiles);
  int wgs = 0;
  for (int p = 0; p < 4; ++p) {
    int n = std::max(1, (int)((int64_t)blocks * cost[p] / total));
    n = (int)std::min<int64_t>(n, ntiles);
    g.tiles_per_wg[p] = crn_cdiv(ntiles, n);
    n = crn_cdiv(ntiles, g.tiles_per_wg[p]);
    wgs += n;
    g.wg_end[p] = wgs;
  }
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)convt_par_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsWg) == hipSuccess;
  }();
  if (!attr) return CRN_EINVAL;
  hipLaunchKernelGGL(convt_par_wgrad_kernel, dim3((unsigned)wgs), dim3(kT), kLdsWg, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

// Two classes (h7): the resident-weights kernels (see convt_res_kernel).  wimg: conv_geometry.convt_res_fwd_table / _dgrad_table.
extern "C" int crn_convt_s2k7_c2_fwd_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                                         const void* wimg, const float* bias, float* y, int64_t y_sB, int64_t y_sC, int Cout,
                                         crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !wimg || !y || B < 1 || Cout != 2) return CRN_EINVAL;
  if (D % TD || H % TH || W % TW) return CRN_EINVAL;
  if ((int64_t)16 * D * H * W >= ((int64_t)1 << 29)) return CRN_EINVAL;
  if ((((uintptr_t)x) & 15) || (x_sB & 3) || (((uintptr_t)y) & 15) || (y_sB & 3) || (y_sC & 3)) return CRN_EINVAL;
  CtGeom g{};
  g.x = x; g.x_sB = x_sB; g.B = B; g.D = D; g.H = H; g.W = W;
  if (tr && tr->scale) { g.scale = tr->scale; g.shift = tr->shift; g.pre_relu = tr->pre_relu; g.post_relu = tr->post_relu; }
  g.wimg = wimg; g.bias = bias; g.y = y; g.y_sB = y_sB; g.y_sC = y_sC; g.Cout = Cout;
  g.tilesD = D / TD; g.tilesH = H / TH; g.tilesW = W / TW;
  const int64_t tiles = (int64_t)B * g.tilesD * g.tilesH * g.tilesW;
  if (tiles > 0x7fffffff) return CRN_EINVAL;
  static const int kWgs = getenv("CRN_CT_RES_WGS") ? std::max(1, atoi(getenv("CRN_CT_RES_WGS"))) : 256;
  static const bool kXcd = getenv("CRN_CT_RES_XCD") == nullptr || atoi(getenv("CRN_CT_RES_XCD")) != 0;
  g.xcd = kXcd ? 1 : 0;
  const int per = crn_cdiv(tiles, std::min<int64_t>(tiles, kWgs)), wgs = crn_cdiv(tiles, per);
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)convt_res_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsRes) == hipSuccess;
  }();
  if (!attr) return CRN_EINVAL;
  hipLaunchKernelGGL(convt_res_kernel<false>, dim3((unsigned)wgs), dim3(kT), kLdsRes, (hipStream_t)stream, g, (int)tiles, per);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_convt_s2k7_c2_dgrad_bf3(const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, int B, int D, int H, int W,
                                           const void* wimg, float* dx, int64_t dx_sB, int accumulate, crnStream stream) {
  CRN_ENTRY(stream);
  if (!dy || !wimg || !dx || B < 1 || Cout < 1 || Cout > 2) return CRN_EINVAL;
  if (D % TD || H % TH || W % TW) return CRN_EINVAL;
  if ((int64_t)Cout * dy_sC >= ((int64_t)1 << 29)) return CRN_EINVAL;
  if ((((uintptr_t)dy) & 15) || (dy_sB & 3) || (dy_sC & 3) || (((uintptr_t)dx) & 15) || (dx_sB & 3)) return CRN_EINVAL;
  CtGeom g{};
  g.x = dx; g.x_sB = dx_sB; g.B = B; g.D = D; g.H = H; g.W = W;
  g.wimg = wimg; g.y = const_cast<float*>(dy); g.y_sB = dy_sB; g.y_sC = dy_sC; g.Cout = Cout; g.accumulate = accumulate ? 1 : 0;
  g.tilesD = D / TD; g.tilesH = H / TH; g.tilesW = W / TW;
  const int64_t tiles = (int64_t)B * g.tilesD * g.tilesH * g.tilesW;
  if (tiles > 0x7fffffff) return CRN_EINVAL;
  static const int kWgs = getenv("CRN_CT_RES_WGS") ? std::max(1, atoi(getenv("CRN_CT_RES_WGS"))) : 256;
  static const bool kXcd = getenv("CRN_CT_RES_XCD") == nullptr || atoi(getenv("CRN_CT_RES_XCD")) != 0;
  g.xcd = kXcd ? 1 : 0;
  const int per = crn_cdiv(tiles, std::min<int64_t>(tiles, kWgs)), wgs = crn_cdiv(tiles, per);
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)convt_res_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsRes) == hipSuccess;
  }();
  if (!attr) return CRN_EINVAL;
  hipLaunchKernelGGL(convt_res_kernel<true>, dim3((unsigned)wgs), dim3(kT), kLdsRes, (hipStream_t)stream, g, (int)tiles, per);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
