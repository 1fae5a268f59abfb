// Split-bf16 ("bf16x3") MFMA convolution engine for the big decoder layers (gfx950 / CDNA4).
//
// Same operation and the same operands as crn_conv_fwd (conv_igemm.hip): one stride-1 window correlation
// over a logical NCDHW view with packed fp32 weights [Cin][taps][Npad]; it serves the forward pass and the
// data gradient of Conv3d k5 / ConvTranspose3d k7 s2 of decoder stages 4-6
// (reconstruction_decoder.py:72-95), where 70 % of the step's FLOPs are.
//
// Why: the fp32 MFMA (v_mfma_f32_16x16x4_f32) peaks at 157 TF/s; the bf16 MFMA at 2.5 PF/s.  Every fp32
// operand is split into two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi); together 16 mantissa
// bits), and a product is three MFMAs with fp32 accumulation:  hi*hi + hi*lo + lo*hi  (the dropped lo*lo
// term is 2^-16 relative).  Measured inner loop (tools/mfma_bf3_loop.hip): 540-650 TF/s fp32-equivalent.
// The fp32 path stays the parity default; this path is selected per layer by the host
// (Engine(decoder_math="bf16x3")) and is measured against it in tests/test_kernels_gpu.py.
//
// Layout (one workgroup = 512 threads = 8 waves, 1 per CU, 2 waves per SIMD):
//  * tile = 4 x 8 x 16 output positions = 32 sub-tiles of 16 W positions, 4 per wave (MSUB);
//    N block = NSUB * 16 output channels.
//  * K of v_mfma_f32_16x16x32_bf16 = 4 window taps (lane group kk = lane >> 4) x 8 input channels
//    (the 8 bf16 of a lane's operand register).  The in-plane taps (zh, zw) of the window are flattened and
//    cut into groups of 4 (5x5 -> 7 groups, 4x4 -> 4 groups); the lane's LDS address is its position plus
//    the offset of ITS tap, so a window tap is still an LDS offset, never a gather.
//  * LDS: input patch of one chunk of 8 channels as [position][8 x bf16] = 16 B per position, a hi plane
//    and a lo plane (one ds_read_b128 per operand); weights of one (chunk, zd) slab as
//    [tap][n][8 x bf16] hi / lo.  Both are split while they are staged (fp32 in HBM, no pre-pass), with the
//    BatchRenorm-apply + ReLU of the producer fused into the patch staging like in the fp32 engine.
//  * register-staged pipeline: global loads of the next slab / next chunk's patch fly under the MFMAs.
#include "conv_kernels.h"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace crnk;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512, kMSUB = 4, kCK = 8;
constexpr int kNUX = 2;          // patch units (2 positions x 8 channels) per thread
constexpr int kTabC = 512;       // channel tables
constexpr int kMaxTapSlots = 32; // NG * 4

struct Bf3Geom {
  crnView x, y;
  crnInTransform tr;
  const float* w;
  const float* bias;
  int Npad, bias_sB;
  int kd, kh, kw, pd, ph, pw, T, KHW;
  int mw, mh, nsw, nsh;    // sub-tile = mh x mw positions (16 MFMA rows), tile = TD x (nsh*mh) x (nsw*mw), 32 sub-tiles
  int TD, TH, TW;
  int PD, PH, PW, PHW, NP; // patch dims (positions), PH*PW, PD*PH*PW
  int pw2, nunits;         // position pairs per row, staging units
  int NG;                  // groups of 4 in-plane taps
  int tilesD, tilesH, tilesW;
  int nchunks, chunks_per_split;
  int ctab;                // entries of the per-channel LDS tables (multiple of 16)
  int mode;                // 0 store, 1 accumulate, 3 split-K partial sums to a dense scratch [split][b][n][pos]
                           // (+ reduction launch), 4 the same, summed by the last workgroup of each tile
  crnView yreal; int accumulate_real; int* counters;      // mode 4
  int lead;
  int vec_store;
  int n_groups, c_groups;
  signed char n_box[8][6], c_box[8][6];
  unsigned magic_pw2, magic_PH, magic_kw;
  const void* wslab;       // weights pre-arranged as slab images (crn_bf3_operands, slab order), or nullptr: w is staged
  int UP;                  // wave-specialised kernel: staging units per patch plane (PH * pw2)
  unsigned magic_UP;
  int dbg;
  int rowskip;             // 4 x 4 window planes with tap boxes: a tap group = a window row, rows outside the box are skipped
  int half_last;           // the last chunk has <= 4 real channels: 8 taps x 4 channels per MFMA (conv_bf3_kernel, plane_half)
  long long* stamps;       // tuning aid (CRN_BF3_STAMPS=1): shader-clock stamps of workgroup 0, 4 per staging step
  // fused sums of the BatchRenorm backward whose output gradient this launch writes (crn_conv_fwd_bf3_slabs_bnbwd)
  const float* bn_x; int64_t bn_sB, bn_S; const float* bn_saved; int bn_pre_relu;
  double* bn_ws; float* bn_dsum; int bn_ndsum;
};

// Epilogue of a data gradient in front of a BatchRenorm backward: s1[ns] / s2[ns] hold this lane's share of sum(g) and
// sum(g * xn) of channel n0 + ns * 16 + i16 (16 values: 4 sub-tiles x 4 W positions).  Lanes kk = 0..3 of a channel are
// added by shuffles, the 8 MFMA waves through LDS (the first bytes of the workgroup's LDS: the channel tables are dead),
// in double from there on; workgroup slot blockIdx.x of [C][gridDim.x][2].  Every wave of the workgroup that is still
// running must call this (two barriers).
template <int NSUB>
__device__ __forceinline__ void bn_bwd_sums_store(const Bf3Geom& g, char* smem, float (&s1)[NSUB], float (&s2)[NSUB],
                                                  int wave, int kk, int i16, int n0, int tid) {
  constexpr int NB = NSUB * 16;
  float* red = reinterpret_cast<float*>(smem);                 // [8 waves][NB][2]
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    s1[ns] += __shfl_xor(s1[ns], 16); s1[ns] += __shfl_xor(s1[ns], 32);
    s2[ns] += __shfl_xor(s2[ns], 16); s2[ns] += __shfl_xor(s2[ns], 32);
  }
  __syncthreads();                                             // every wave is past its last read of the LDS images
  if (kk == 0 && wave < 8) {
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      red[((wave * NB) + ns * 16 + i16) * 2 + 0] = s1[ns];
      red[((wave * NB) + ns * 16 + i16) * 2 + 1] = s2[ns];
    }
  }
  __syncthreads();
  if (tid < NB) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += (double)red[(w * NB + tid) * 2]; b += (double)red[(w * NB + tid) * 2 + 1]; }
    const int n = n0 + tid;
    if (n < g.y.C) {
      double* o = g.bn_ws + ((int64_t)n * gridDim.x + blockIdx.x) * 2;
      o[0] = a; o[1] = b;
    }
  }
  if (g.bn_dsum && blockIdx.x == 0 && blockIdx.y == 0 && tid < g.bn_ndsum) g.bn_dsum[tid] = 0.f;
}

__device__ __forceinline__ void bload2(f32x2& dst, const crn_rsrc& rs, unsigned byte_off) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rs));
}
template <typename VT, int A, int B>
__device__ __forceinline__ void wait_loads2d(VT (&v)[A][B]) {
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < B; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i][j]));
}

template <int N, typename VT, int A, int B>
__device__ __forceinline__ void wait_loads2d_n(VT (&v)[A][B]) {      // ... until at most N (newer) loads are outstanding
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < B; ++j) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[i][j]) : "n"(N));
}

template <int N, typename VT, int A>
__device__ __forceinline__ void wait_loads1d_n(VT (&v)[A]) {
#pragma unroll
  for (int i = 0; i < A; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[i]) : "n"(N));
}

// x = hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}


// The three products of one accumulator -- hi*hi, hi*lo, lo*hi -- as ONE block of three adjacent MFMAs.  Left to the compiler they
// come out as two adjacent MFMAs and a third one behind LDS reads and address arithmetic; three dependent MFMAs with idle cycles
// between them make VALU results of OTHER waves on the SIMD go missing (the ray scatter beside these kernels, DESIGN section 3e,
// tools/mfma_neighbour.py: crn_mfma_probe mode 48 does it in 29 of 30 runs, mode 32 -- this shape -- never).  Same additions in the
// same order: results are bit-identical.
__device__ __forceinline__ void mfma3(f32x4& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
               : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl));
}

template <int XM> struct XLoad;
template <> struct XLoad<1> {            // unit-stride view: 2 consecutive positions = one 8-byte load
  typedef f32x2 T;
  static __device__ __forceinline__ void load(T& d, const crn_rsrc& rs, unsigned off) { bload2(d, rs, off); }
  static __device__ __forceinline__ float e0(const T& v) { return v[0]; }
  static __device__ __forceinline__ float e1(const T& v) { return v[1]; }
};
template <> struct XLoad<2> {            // stride-2 (space-to-depth) view: elements 0 and 2 of a 12-byte load
  typedef f32x3 T;
  static __device__ __forceinline__ void load(T& d, const crn_rsrc& rs, unsigned off) { crn_bload3(d, rs, off); }
  static __device__ __forceinline__ float e0(const T& v) { return v[0]; }
  static __device__ __forceinline__ float e1(const T& v) { return v[2]; }
};

// NG = groups of 4 in-plane taps (7: 5x5 window, 4: 4x4 window): compile time, so that the group loop is
// straight-line code (LDS reads of the next group issue under the MFMAs of the current one) and every
// staging loop has exactly the trip count the layer needs.
// ZS = window planes (zd) per weight slab: the slab of one chunk is staged ZS planes at a time; with ZS = the
// whole window depth a chunk is ONE staging step (two barriers) instead of one per plane.
// WS: the weights come pre-split and pre-arranged in the order of the LDS slab ([chunk][zd][tap slot][n] entries of
// 8 hi + 8 lo bf16, crn_bf3_operands): staging a slab is then two 16-byte loads and two LDS writes per item instead of
// eight dword loads, eight splits and two writes (per staging step: ~930 -> ~300 cycles of load issue and ~400 -> ~150
// of commit, CRN_BF3_STAMPS; both sit outside the MFMA phase).
template <int NSUB, int XM, int NG, int ZS, bool WS, bool HALF>
__device__ __forceinline__ void conv_bf3_body(const Bf3Geom& g) {
  crn_kernarg_touch(g);
  constexpr int NB = NSUB * 16;
  constexpr int kSlab = ZS * NG * 4 * NB;                                 // weight items (16-byte units) per slab
  constexpr int kNWI = (kSlab + kThreads - 1) / kThreads;                 // ... per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // channel tables sized for this layer (ctab = channels rounded up to 16): with 28..32 channels the k5 / N 16
  // configuration stays under 80 KiB of LDS and two workgroups share a CU (measured: stage_6.c1 forward 391 us
  // with two, 509 us with one)
  unsigned* choff = reinterpret_cast<unsigned*>(smem);                    // [ctab]
  float* tscale = reinterpret_cast<float*>(smem + g.ctab * 4);            // [ctab]
  float* tshift = reinterpret_cast<float*>(smem + 2 * g.ctab * 4);        // [ctab]
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + 3 * g.ctab * 4);
  bf16x8* Alo = Ahi + g.NP;
  bf16x8* Bhi = Alo + g.NP;                                               // [ZS][NG*4][NB]
  bf16x8* Blo = Bhi + kSlab;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  typedef typename XLoad<XM>::T XT;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * g.TD, h0 = thi * g.TH, w0 = twi * g.TW;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int cbeg = split * g.chunks_per_split, cend = min(cbeg + g.chunks_per_split, g.nchunks);

  for (int c = tid; c < g.ctab; c += kThreads) {
    const int cc = min(c, g.x.C - 1);
    choff[c] = g.x.chan_off ? (unsigned)g.x.chan_off[cc] : (unsigned)cc * (unsigned)g.x.sC;
    tscale[c] = g.tr.scale ? g.tr.scale[cc] : 1.f;
    tshift[c] = g.tr.scale ? g.tr.shift[cc] : 0.f;
  }

  int toff[NG];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    const int t = gq * 4 + kk < g.KHW ? gq * 4 + kk : 0;   // slots past the window carry zero weights
    const int zh = mdiv(t, g.magic_kw), zw = t - zh * g.kw;
    toff[gq] = zh * g.PW + zw;
  }
  const int ri = i16 / g.mw, rj = i16 - ri * g.mw;
  int pa[kMSUB];
#pragma unroll
  for (int ms = 0; ms < kMSUB; ++ms) {
    int s_ = wave * kMSUB + ms;
    const int sw = s_ % g.nsw; s_ /= g.nsw;
    const int sh = s_ % g.nsh, sd = s_ / g.nsh;
    pa[ms] = (sd * g.PH + sh * g.mh + ri) * g.PW + sw * g.mw + rj + g.lead;
  }
  // (HALF: the launcher guarantees 16-wide sub-tiles on a row pitch of 20 -- a wave's four sub-tiles are four consecutive rows,
  // so three of the four base addresses are immediates of the LDS reads: the registers this instantiation has to give back)
  auto PA = [&](int ms) -> unsigned { if constexpr (HALF) return (unsigned)pa[0] + (unsigned)(ms * 20); else return (unsigned)pa[ms]; };
  f32x4 acc[kMSUB][NSUB];
#pragma unroll
  for (int ms = 0; ms < kMSUB; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
  const crn_rsrc wrs = make_rsrc(g.w);
  const crn_rsrc wsrs = make_rsrc(reinterpret_cast<const float*>(g.wslab));
  XT pv[kNUX][kCK];
  float wv[WS ? 1 : kNWI][kCK];
  f32x4 wq[WS ? kNWI : 1][2];
  unsigned inmask = 0;

  // ---- patch staging: unit u = (plane row, position pair); all 8 channels of the chunk ----
  auto unit_of = [&](int j, int& pos, unsigned& sp, bool& in) -> bool {
    int u = tid + j * kThreads;
    asm volatile("" : "+v"(u));
    const int row = mdiv(u, g.magic_pw2), pp = u - row * g.pw2;
    const int pdz = mdiv(row, g.magic_PH), phy = row - pdz * g.PH;
    const int gd = d0 + pdz - g.pd, gh = h0 + phy - g.ph, gw = w0 + 2 * pp - g.pw;
    in = (unsigned)gd < (unsigned)g.x.D && (unsigned)gh < (unsigned)g.x.H && (unsigned)gw < (unsigned)g.x.W;
    sp = (unsigned)gd * (unsigned)g.x.sD + (unsigned)gh * (unsigned)g.x.sH + (unsigned)gw * (unsigned)(XM == 2 ? 2 : 1);
    pos = row * g.PW + 2 * pp;
    return u < g.nunits;
  };
  auto patch_issue = [&](int c0) {
    inmask = 0;
    // the chunk's eight channel offsets with two 16-byte LDS reads (one dword read per load made every load wait for
    // its own LDS round trip: 16 of them in a row per staging step)
    // (NSUB 1 only: in the wider variants the 16 extra live registers cross 128 and cost the second workgroup per CU --
    // stage_5.t1 forward 200 -> 232 us)
    u32x4 co0, co1;
    if constexpr (NSUB == 1) { co0 = *reinterpret_cast<const u32x4*>(choff + c0); co1 = *reinterpret_cast<const u32x4*>(choff + c0 + 4); }
#pragma unroll
    for (int j = 0; j < kNUX; ++j) {
      int pos; unsigned sp; bool in;
      const bool valid = unit_of(j, pos, sp, in);
      const bool ld = valid && in;
      if (ld) inmask |= 1u << j;
      if constexpr (NSUB == 1) {
        unsigned off[kCK];
#pragma unroll
        for (int cl = 0; cl < kCK; ++cl)
          off[cl] = (ld && c0 + cl < g.x.C) ? ((cl < 4 ? co0[cl & 3] : co1[cl & 3]) + sp) * 4u : 0x80000000u;
#pragma unroll
        for (int cl = 0; cl < kCK; ++cl) XLoad<XM>::load(pv[j][cl], xrs, off[cl]);
      } else {
#pragma unroll
        for (int cl = 0; cl < kCK; ++cl) {
          const unsigned off = (ld && c0 + cl < g.x.C) ? (choff[c0 + cl] + sp) * 4u : 0x80000000u;
          XLoad<XM>::load(pv[j][cl], xrs, off);
        }
      }
    }
  };
  auto patch_commit = [&](int c0) {
    f32x4 s0, s1, h0, h1;
    if (NSUB == 1 && g.tr.scale) {
      s0 = *reinterpret_cast<const f32x4*>(tscale + c0); s1 = *reinterpret_cast<const f32x4*>(tscale + c0 + 4);
      h0 = *reinterpret_cast<const f32x4*>(tshift + c0); h1 = *reinterpret_cast<const f32x4*>(tshift + c0 + 4);
    }
#pragma unroll
    for (int j = 0; j < kNUX; ++j) {
      if (j * kThreads < g.nunits) {                     // block-uniform
        int pos; unsigned sp; bool in;
        const bool valid = unit_of(j, pos, sp, in);
        if (valid) {
          float v0[kCK], v1[kCK];
#pragma unroll
          for (int cl = 0; cl < kCK; ++cl) { v0[cl] = XLoad<XM>::e0(pv[j][cl]); v1[cl] = XLoad<XM>::e1(pv[j][cl]); }
          if (g.tr.scale && ((inmask >> j) & 1u)) {        // zero padding stays zero
#pragma unroll
            for (int cl = 0; cl < kCK; ++cl) {
              if (c0 + cl < g.x.C) {
                const float sc = NSUB == 1 ? (cl < 4 ? s0[cl & 3] : s1[cl & 3]) : tscale[c0 + cl];
                const float sh = NSUB == 1 ? (cl < 4 ? h0[cl & 3] : h1[cl & 3]) : tshift[c0 + cl];
                float a = v0[cl], c = v1[cl];
                if (g.tr.pre_relu) { a = fmaxf(a, 0.f); c = fmaxf(c, 0.f); }
                a = a * sc + sh; c = c * sc + sh;
                if (g.tr.post_relu) { a = fmaxf(a, 0.f); c = fmaxf(c, 0.f); }
                v0[cl] = a; v1[cl] = c;
              }
            }
          }
          bf16x8 h0v, l0v, h1v, l1v;
          split8(v0, h0v, l0v);
          split8(v1, h1v, l1v);
          Ahi[pos] = h0v; Ahi[pos + 1] = h1v;
          Alo[pos] = l0v; Alo[pos + 1] = l1v;
        }
      }
    }
  };
  // ---- weight staging: item = (plane zs of the slab, in-plane tap slot tp, column n), all 8 channels ----
  auto weights_issue = [&](int c0, int zd0) {
    if constexpr (WS) {
      static_assert(!WS || ZS == 1, "slab operands are one window plane per slab");
      const unsigned sbase = (unsigned)((c0 / kCK) * g.kd + zd0) * (unsigned)(NG * 4);
#pragma unroll
      for (int j = 0; j < kNWI; ++j) {
        int it = tid + j * kThreads;
        asm volatile("" : "+v"(it));
        const int n = it & (NB - 1), tp = it / NB;
        const bool ok = it < kSlab && n0 + n < g.Npad;
        const unsigned off = ok ? ((sbase + (unsigned)tp) * (unsigned)g.Npad + (unsigned)(n0 + n)) * 32u : 0x80000000u;
        crn_bload4(wq[j][0], wsrs, off);
        crn_bload4(wq[j][1], wsrs, ok ? off + 16u : 0x80000000u);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < kNWI; ++j) {
      int it = tid + j * kThreads;
      asm volatile("" : "+v"(it));
      const int n = it & (NB - 1), zt = it / NB;              // zt = zs * NG*4 + tp
      const int zs = ZS == 1 ? 0 : zt / (NG * 4), tp = zt - zs * (NG * 4);
      const bool ok = it < kSlab && tp < g.KHW && zd0 + zs < g.kd && n0 + n < g.Npad;
      const unsigned base = (unsigned)(((zd0 + zs) * g.KHW + tp) * g.Npad + n0 + n);
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl) {
        const unsigned off = (ok && c0 + cl < g.x.C) ? ((unsigned)(c0 + cl) * (unsigned)(g.T * g.Npad) + base) * 4u : 0x80000000u;
        crn_bload(wv[j][cl], wrs, off);
      }
    }
  };
  auto weights_commit = [&]() {
#pragma unroll
    for (int j = 0; j < kNWI; ++j) {
      if (j * kThreads < kSlab) {
        int it = tid + j * kThreads;
        asm volatile("" : "+v"(it));
        if (it < kSlab) {
          bf16x8 h, l;
          if constexpr (WS) { h = __builtin_bit_cast(bf16x8, wq[j][0]); l = __builtin_bit_cast(bf16x8, wq[j][1]); }
          else split8(wv[j], h, l);
          Bhi[it] = h;                                   // [zs][tp][n] with n = ns*16 + n16: the fragment order
          Blo[it] = l;
        }
      }
    }
  };

  const auto nbox = box_union(g.n_box, g.n_groups, g.y.C, n0, min(n0 + NB, g.y.C) - 1, g.kd, g.kh, g.kw);
  // zd range of a chunk that can hold non-zero weights (tap boxes of transposed convolutions)
  // (hr: the window ROWS zh of that range that can hold non-zero weights, h0 | h1 << 8 -- for 4 x 4 window planes a row is one
  // group of four taps, and a group outside the box multiplies zeros only: `plane` skips it, g.rowskip)
  auto chunk_zrange = [&](int c0, int& z0, int& z1, int& hr) {
    const TapBox tb = box_intersect(nbox, box_union(g.c_box, g.c_groups, g.x.C, c0, min(c0 + kCK, g.x.C) - 1,
                                                    g.kd, g.kh, g.kw));
    z0 = tb.d0; z1 = tb.d1;
    if (tb.h1 <= tb.h0 || tb.w1 <= tb.w0) z1 = z0;
    hr = tb.h0 | (tb.h1 << 8);
  };
  // first (chunk, slab) step with work at or after `chunk`
  auto first_step = [&](int chunk, int& zd, int& z1, int& hr) -> int {
    for (; chunk < cend; ++chunk) {
      int z0;
      chunk_zrange(chunk * kCK, z0, z1, hr);
      if (z1 > z0) { zd = ZS == 1 ? z0 : 0; return chunk; }
    }
    return cend;
  };

  __syncthreads();                                       // tables are in LDS
  int step_no = 0;
  const bool stamp = g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
  auto mark = [&](int i) { if (stamp && step_no < 24) g.stamps[step_no * 8 + i] = (long long)__builtin_amdgcn_s_memtime(); };
  int zd = 0, zend = 0, hrow = 0;
  int chunk = first_step(cbeg, zd, zend, hrow);
  bool first = true;
  bool fresh = true;                                     // the patch of `chunk` is in registers, not yet in LDS
  if (chunk < cend) {
    patch_issue(chunk * kCK);
    weights_issue(chunk * kCK, zd);
  }
  // (the steps of a last chunk that multiplies 8 taps x 4 channels -- plane_half below -- are a second copy of the loop: a
  // branch inside it would be a control-flow merge under the staging registers, +8 VGPRs and the second workgroup of the CU)
  auto run_steps = [&](auto half_tag, int cstop) {
  constexpr bool kHalf = decltype(half_tag)::value;
  while (chunk < cstop) {
    const int c0 = chunk * kCK;
    // next step
    int nchunk = chunk, nzd = zd + ZS, nzend = zend, nhrow = hrow;
    if (nzd >= zend) nchunk = first_step(chunk + 1, nzd, nzend, nhrow);
    mark(0);
    wait_loads2d(pv);
    if constexpr (WS) wait_loads2d(wq); else wait_loads2d(wv);
    mark(1);
    __syncthreads();                                     // every wave is done reading the LDS of the previous step
    mark(2);
    if (g.dbg < 3 || first) {
      if (fresh) patch_commit(c0);
      weights_commit();
    }
    first = false;
    mark(3);
    __syncthreads();
    mark(4);
    if (nchunk < cend && g.dbg != 2) {
      if (nchunk != chunk) patch_issue(nchunk * kCK);    // flies under this slab's MFMAs
      weights_issue(nchunk * kCK, nzd);
    }
    mark(5);
    if (g.dbg != 1) {
      // one window plane: every group runs (weights outside a tap box are zero): no control flow between the
      // groups, so the compiler keeps the next group's LDS reads in flight under this group's MFMAs
      auto plane = [&](int z, const bf16x8* bh0, const bf16x8* bl0) {
        const int zoff = z * g.PHW;
        const int rlo = hrow & 255, rhi = hrow >> 8;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          if (NG == 4 && g.rowskip && (gq < rlo || gq >= rhi)) continue;   // (workgroup-uniform: a scalar branch)
          const int off = toff[gq] + zoff;
          bf16x8 bh[NSUB], bl[NSUB], ah[kMSUB], al[kMSUB];
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            bh[ns] = bh0[(gq * 4 + kk) * NB + ns * 16 + i16];
            bl[ns] = bl0[(gq * 4 + kk) * NB + ns * 16 + i16];
          }
#pragma unroll
          for (int ms = 0; ms < kMSUB; ++ms) { ah[ms] = Ahi[PA(ms) + (unsigned)off]; al[ms] = Alo[PA(ms) + (unsigned)off]; }
#pragma unroll
          for (int ms = 0; ms < kMSUB; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
              mfma3(acc[ms][ns], ah[ms], al[ms], bh[ns], bl[ns]);
            }
        }
      };
      // HALF CHUNK.  A last chunk with at most 4 real channels (stage_6.c1: 28 = 3 x 8 + 4) multiplies 8 TAPS x 4 channels per
      // MFMA instead of 4 taps x 8 channels of which half are padding: lane group kk owns taps 8 gq + 2 kk and + 1, its operand
      // registers are the low halves (channels 0-3: 8 bytes) of the two taps' entries, for the patch and for the weight slab
      // alike -- two ds_read_b64 per fragment, the same LDS images.  ceil(KHW / 8) groups per plane instead of ceil(KHW / 4):
      // 4 instead of 7 for a 5 x 5 plane, 125 instead of 140 MFMA triples per sub-tile for stage_6.c1 (the matrix pipe of
      // this part is power-limited at ~1.4 PFLOP/s under this load -- a lone wave per SIMD issues one MFMA per 28 cycles with
      // or without its LDS reads, round 4 -- so executed MFMAs, not bubbles, are what is left to remove).
      auto plane_half = [&](int z, const bf16x8* bh0, const bf16x8* bl0) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        constexpr int NGH = (NG * 4 + 7) / 8;
        const int zoff = z * g.PHW;
        auto lo8 = [](const bf16x8* p) { return *reinterpret_cast<const unsigned long long*>(p); };
        int kk2 = kk;
        asm volatile("" : "+v"(kk2));                           // (the tap arithmetic below stays inside the step: 5 of a tile's 20)
#pragma unroll
        for (int gq = 0; gq < NGH; ++gq) {
          // (slots KHW .. 4 NG - 1 of the slab hold zero weights; a slot past the slab reads the last of them)
          const int t0 = min(gq * 8 + 2 * kk2, NG * 4 - 1), t1 = min(gq * 8 + 2 * kk2 + 1, NG * 4 - 1);
          const int u0 = t0 < g.KHW ? t0 : 0, u1 = t1 < g.KHW ? t1 : 0;
          const int zh0 = mdiv(u0, g.magic_kw), zh1 = mdiv(u1, g.magic_kw);
          const int o0 = zh0 * g.PW + (u0 - zh0 * g.kw) + zoff, o1 = zh1 * g.PW + (u1 - zh1 * g.kw) + zoff;
          const int s0 = t0 * NB + i16, s1 = t1 * NB + i16;
          bf16x8 bh[NSUB], bl[NSUB];
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            bh[ns] = __builtin_bit_cast(bf16x8, (u64x2){lo8(bh0 + s0 + ns * 16), lo8(bh0 + s1 + ns * 16)});
            bl[ns] = __builtin_bit_cast(bf16x8, (u64x2){lo8(bl0 + s0 + ns * 16), lo8(bl0 + s1 + ns * 16)});
          }
          constexpr int MQ = 1;                                  // sub-tiles per pass: the kernel has to stay at 128 registers
#pragma unroll
          for (int mh = 0; mh < kMSUB; mh += MQ) {
            bf16x8 ah[MQ], al[MQ];
#pragma unroll
            for (int m = 0; m < MQ; ++m) {
              ah[m] = __builtin_bit_cast(bf16x8, (u64x2){lo8(Ahi + (PA(mh + m) + (unsigned)o0)), lo8(Ahi + (PA(mh + m) + (unsigned)o1))});
              al[m] = __builtin_bit_cast(bf16x8, (u64x2){lo8(Alo + (PA(mh + m) + (unsigned)o0)), lo8(Alo + (PA(mh + m) + (unsigned)o1))});
            }
#pragma unroll
            for (int m = 0; m < MQ; ++m)
#pragma unroll
              for (int ns = 0; ns < NSUB; ++ns) {
                mfma3(acc[mh + m][ns], ah[m], al[m], bh[ns], bl[ns]);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      if constexpr (ZS == 1) {
        if constexpr (kHalf) plane_half(zd, Bhi, Blo);
        else plane(zd, Bhi, Blo);
      } else {
        int z0, z1, hr_;
        chunk_zrange(c0, z0, z1, hr_);
        const int zlo = max(zd, z0), zhi = min(zd + ZS, zend);
#pragma unroll 1
        for (int z = zlo; z < zhi; ++z) {
          int zz = z;
          asm volatile("" : "+v"(zz));                    // keep the planes a real loop (registers: one plane in flight)
          plane(zz, Bhi + (zz - zd) * (NG * 4 * NB), Blo + (zz - zd) * (NG * 4 * NB));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    mark(6);
    ++step_no;
    fresh = nchunk != chunk;
    chunk = nchunk; zd = nzd; zend = nzend; hrow = nhrow;
  }
  };
  const int chalf = (HALF && cend * kCK >= g.x.C) ? cend - 1 : cend;
  run_steps(std::false_type(), chalf);
  if constexpr (HALF) run_steps(std::true_type(), cend);

  // epilogue: D row = kk*4 + r = position (kk*4 + r) of the mh x mw sub-tile, col = i16 = channel.
  // mode 3: split-K partial sums go to a dense scratch tensor [split][b][n][pos]; a reduction launch adds them up
  float* yb = g.y.base + (int64_t)(g.mode >= 3 ? split * g.x.B + b : b) * g.y.sB;
  float bn_s1[NSUB], bn_s2[NSUB], bn_mu[NSUB], bn_rstd[NSUB];
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    bn_s1[ns] = bn_s2[ns] = 0.f;
    const int n = n0 + ns * 16 + i16;
    const bool on = g.bn_x && n < g.y.C;
    bn_mu[ns] = on ? g.bn_saved[n] : 0.f; bn_rstd[ns] = on ? g.bn_saved[g.y.C + n] : 0.f;
  }
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    const int n = n0 + ns * 16 + i16;
    if (n >= g.y.C) continue;
    const int64_t co = view_chan(g.y, n);
    const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
#pragma unroll
    for (int ms = 0; ms < kMSUB; ++ms) {
      int s_ = wave * kMSUB + ms;
      const int sw = s_ % g.nsw; s_ /= g.nsw;
      const int sh = s_ % g.nsh, sd = s_ / g.nsh;
      const int p0 = kk * 4;                                        // first of this lane's 4 rows
      const int od = d0 + sd, oh = h0 + sh * g.mh + p0 / g.mw, ow = w0 + sw * g.mw + p0 % g.mw;
      if (od >= g.y.D || oh >= g.y.H || ow >= g.y.W) continue;    // (mw >= 4: the 4 rows are 4 consecutive W positions)
      float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
      if (g.vec_store) {
        f32x4 v = acc[ms][ns] + bsv;
        if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
        if (g.bn_x) {                            // (fused only with vector stores: dense y, dense x of the same shape)
          const f32x4 xv = *reinterpret_cast<const f32x4*>(g.bn_x + (int64_t)b * g.bn_sB + (int64_t)n * g.bn_S +
                                                            ((int64_t)od * g.y.H + oh) * g.y.W + ow);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xr = g.bn_pre_relu ? fmaxf(xv[r], 0.f) : xv[r];
            bn_s1[ns] += v[r];
            bn_s2[ns] += v[r] * ((xr - bn_mu[ns]) * bn_rstd[ns]);
          }
        } else if (g.bn_ws) {                    // forward statistics of the norm BEHIND this conv: sum(y), sum(y^2) (of max(y, 0))
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yr = g.bn_pre_relu ? fmaxf(v[r], 0.f) : v[r];
            bn_s1[ns] += yr;
            bn_s2[ns] += yr * yr;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ow + r < g.y.W) {
            float* d = dst + (int64_t)r * g.y.sW;
            const float v = acc[ms][ns][r] + bsv;
            *d = g.mode == 1 ? *d + v : v;
          }
        }
      }
    }
  }
  if (g.bn_ws) bn_bwd_sums_store<NSUB>(g, smem, bn_s1, bn_s2, wave, kk, i16, n0, tid);
  if (g.mode == 4) {      // fused split-K reduction by the last workgroup of the tile (see conv_fwd_kernel, mode 4)
    __threadfence();
    int* s_last = reinterpret_cast<int*>(smem);           // the channel tables are dead by now
    __syncthreads();
    if (tid == 0) {
      const int id = blockIdx.x + gridDim.x * blockIdx.y;
      const int ticket = atomicAdd(g.counters + id, 1);
      const int last = ticket == (int)gridDim.z - 1;
      if (last) g.counters[id] = 0;
      *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    __threadfence();
    const int64_t slab = (int64_t)g.x.B * g.y.sB;
    const int nsplit = gridDim.z;
    const float* sb = g.y.base + (int64_t)b * g.y.sB;
    float* yr = g.yreal.base + (int64_t)b * g.yreal.sB;
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int n = n0 + ns * 16 + i16;
      if (n >= g.y.C) continue;
      const int64_t co = view_chan(g.y, n), cor = view_chan(g.yreal, n);
#pragma unroll
      for (int ms = 0; ms < kMSUB; ++ms) {
        int s_ = wave * kMSUB + ms;
        const int sw = s_ % g.nsw; s_ /= g.nsw;
        const int sh = s_ % g.nsh, sd = s_ / g.nsh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = kk * 4 + r;
          const int od = d0 + sd, oh = h0 + sh * g.mh + p / g.mw, ow = w0 + sw * g.mw + p % g.mw;
          if (od < g.y.D && oh < g.y.H && ow < g.y.W) {
            const float* src = sb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
            float sum = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) sum += __builtin_nontemporal_load(src + sp * slab);
            float* dst = yr + cor + (int64_t)od * g.yreal.sD + (int64_t)oh * g.yreal.sH + (int64_t)ow * g.yreal.sW;
            *dst = g.accumulate_real ? *dst + sum : sum;
          }
        }
      }
    }
  }
}

template <int NSUB, int XM, int NG, int ZS, bool WS>
__global__ __launch_bounds__(kThreads) void conv_bf3_kernel(Bf3Geom g) { conv_bf3_body<NSUB, XM, NG, ZS, WS, false>(g); }
// ... with a half last chunk (plane_half): two workgroups per CU are worth more than the two registers the second copy of the
// step loop costs: its sub-tile addresses are immediates (PA), 123 registers
template <int NSUB, int XM, int NG, int ZS, bool WS>
__global__ __launch_bounds__(kThreads) void conv_bf3_half_kernel(Bf3Geom g) { conv_bf3_body<NSUB, XM, NG, ZS, WS, true>(g); }

template <typename G>
__host__ __device__ __forceinline__ void bf3_chunk_zrange(const G& g, const TapBox& nbox, int c0, int& z0, int& z1) {
  const int chi = (c0 + kCK < g.x.C ? c0 + kCK : g.x.C) - 1;
  const TapBox tb = box_intersect(nbox, box_union(g.c_box, g.c_groups, g.x.C, c0, chi, g.kd, g.kh, g.kw));
  z0 = tb.d0; z1 = tb.d1;
  if (tb.h1 <= tb.h0 || tb.w1 <= tb.w0) z1 = z0;
}

// ------------------------------------ wave-specialised forward / data gradient --------------------------------
// The kernel above spends ~55 % of a staging step outside its MFMA phase, and what it waits for is not work but the
// ~2 us (4-5 k cycles) a global load takes to come back on a busy chip: every wave issues its loads, multiplies for
// ~2.9 k cycles, waits for the rest of the latency, commits, and sits in two barriers (T ~ T_mfma + T_staging even with
// two workgroups per CU; ablations of a first role-split version: producer chain alone 4.9 k cycles per step,
// consumers alone 5.4 k with compiler-scheduled LDS reads -- profiles/r03_ws_ablate.txt).  Here:
//   * ONE workgroup of 12 waves per CU.  Waves 0-7 ("consumers") only multiply: one (chunk, window plane) step after
//     the other, ONE barrier per step, operand fragments of the next tap group read from LDS before the MFMAs of the
//     current one issue (two waves per SIMD cannot hide an LDS round trip per MFMA triple the way four did).  They also
//     own the weights: while it multiplies step n, a wave copies its share of step n+1's slab from the pre-arranged
//     slab image (L2-resident) into the idle one of two slab buffers by LDS-DMA (buffer_load ... lds: no registers,
//     no VALU), and waits for it at the end of the step.
//   * Waves 8-11 ("producers", one per SIMD) own the activations.  The input patch is a RING of 8 window planes
//     ([position][8 ch] bf16 hi / lo images like above); step n = (chunk c, plane zd) reads planes zd .. zd+3 of the
//     chunk's PD = kd + 3.  Plane p of chunk c is written the moment the plane it replaces is dead, on a fixed
//     timetable: during the step f(p) - 3 steps away from the chunk's first, f = 0,1,2,2,3,3,4,5 -- at most two
//     planes = 240 staging units per step, one unit (2 positions x 8 channels) per producer lane.  Its global loads
//     are issued TWO steps before that into one of three register sets, so a load has two whole steps (~6 k cycles)
//     to come back and a producer never waits: wait (counted vmcnt: the two younger sets stay in flight) ->
//     BatchRenorm-apply + ReLU + split -> LDS, ~1 k cycles of a 2.9 k cycle step.
//     Ring slot of plane p of the c-th chunk: (PD * c + p) mod 8 -- p itself for the 5^3 windows (PD = 8), a rotating
//     slot for the 4^3 windows (PD = 7).  That no step reads a slot a concurrent write touches follows from the
//     timetable (bf3_ws_timetable_ok runs it on the host for every launch geometry).
//   * Tap boxes (transposed convolutions): the timetable ignores them -- every chunk stages all its planes -- and the
//     consumers skip the MFMAs of steps whose window plane holds only structural zeros.
// Same products, same summation order as the kernel above: bit-identical results.
constexpr int kWsConsumers = 8, kWsProducers = 4;
constexpr int kWsThreads = (kWsConsumers + kWsProducers) * 64, kWsPL = kWsProducers * 64;
constexpr int kWsR = 8;                                        // planes of the patch ring

// planes whose ring write (commit) happens in iteration T (T = kd*c + f(p)): up to two (chunk-in-split, plane) pairs
struct Bf3WsPlanes { int n, c0, p0, c1, p1; };
template <int KD>
__host__ __device__ __forceinline__ Bf3WsPlanes bf3_ws_planes_at(int T, int nch) {
  Bf3WsPlanes w{0, 0, 0, 0, 0};
  if (T < 0) return w;
  const int q = T / KD, r = T - q * KD;
  int ca = q, pa = 6, cb = -1, pb = 0;                         // (r == 4, KD == 5 only: plane 6)
  if (r == 0) { pa = 0; cb = q - 1; pb = KD + 2; }
  else if (r == 1) pa = 1;
  else if (r == 2) { pa = 2; cb = q; pb = 3; }
  else if (r == 3) { pa = 4; cb = q; pb = 5; }
  const bool va = ca >= 0 && ca < nch, vb = cb >= 0 && cb < nch;
  if (va) { w.c0 = ca; w.p0 = pa; w.n = 1; if (vb) { w.c1 = cb; w.p1 = pb; w.n = 2; } }
  else if (vb) { w.c0 = cb; w.p0 = pb; w.n = 1; }
  return w;
}

typedef __attribute__((address_space(3))) void lds_void;

template <int NSUB, int XM, int NG>
__global__ __launch_bounds__(kWsThreads) void conv_bf3_ws_kernel(Bf3Geom g) {
  crn_kernarg_touch(g);
  constexpr int NB = NSUB * 16;
  constexpr int KD = NG == 7 ? 5 : 4, PD = KD + 3;             // 5^3 / 4^3 windows on 4 x 8 x 16 tiles
  constexpr int kSlab = NG * 4 * NB;                           // weight items (hi 16 B + lo 16 B) per slab
  static_assert(kSlab % 64 == 0, "a slab is a whole number of 64-lane DMA instructions");
  constexpr int kSlabI = kSlab / 64;                           // LDS-DMA instructions per slab half (hi / lo)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* choff = reinterpret_cast<unsigned*>(smem);                    // [ctab]
  float* tscale = reinterpret_cast<float*>(smem + g.ctab * 4);            // [ctab]
  float* tshift = reinterpret_cast<float*>(smem + 2 * g.ctab * 4);        // [ctab]
  int* zrtab = reinterpret_cast<int*>(smem + 3 * g.ctab * 4);              // [64] z range (z0 | z1 << 8) of a chunk
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + 3 * g.ctab * 4 + 256);   // [8][PHW]
  bf16x8* Alo = Ahi + kWsR * g.PHW;
  bf16x8* Bhi = Alo + kWsR * g.PHW;                                       // [2][NG*4][NB]
  bf16x8* Blo = Bhi + 2 * kSlab;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  const bool producer = wave >= kWsConsumers;
  typedef typename XLoad<XM>::T XT;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * g.TD, h0 = thi * g.TH, w0 = twi * g.TW;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int cbeg = split * g.chunks_per_split, cend = min(cbeg + g.chunks_per_split, g.nchunks);
  const int nch = cend - cbeg, nsteps = nch * KD;
  const int niter = (nsteps + 3 + 2) / 3 * 3;        // iterations (= barriers) of both roles: a multiple of three

  for (int c = tid; c < g.ctab; c += kWsThreads) {
    const int cc = min(c, g.x.C - 1);
    choff[c] = g.x.chan_off ? (unsigned)g.x.chan_off[cc] : (unsigned)cc * (unsigned)g.x.sC;
    tscale[c] = g.tr.scale ? (c < g.x.C ? g.tr.scale[cc] : 0.f) : 1.f;      // (channels past C are loaded as 0 and stay 0)
    tshift[c] = g.tr.scale && c < g.x.C ? g.tr.shift[cc] : 0.f;
  }

  if (tid < nch) {       // window planes of each chunk that can hold non-zero weights (tap boxes of transposed convs)
    const TapBox nbox = box_union(g.n_box, g.n_groups, g.y.C, n0, min(n0 + NB, g.y.C) - 1, g.kd, g.kh, g.kw);
    int z0, z1;
    bf3_chunk_zrange(g, nbox, (cbeg + tid) * kCK, z0, z1);
    zrtab[tid] = z0 | (z1 << 8);
  }
  __syncthreads();       // the tables are in LDS (the producers read them before the first barrier of the loops)

  if (producer) {
    // ------------------------------------------------ producers ------------------------------------------------
    // (the producer is the youngest wave of its SIMD and would lose every issue arbitration against the two consumers:
    // its ~300 VALU / LDS / VMEM instructions per step took 4-5 k cycles that way -- more than the step; with priority
    // they slot in between the MFMAs, which issue from their own pipe)
    if (g.dbg != 6) __builtin_amdgcn_s_setprio(3);
    const int ptid = tid - kWsConsumers * 64;
    const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
    XT pv[3][kCK];
    unsigned inm[3] = {0, 0, 0};
    // this lane's unit of an iteration's (up to two) planes: plane pz = ptid / UP, patch row, position pair
    const int pz = ptid >= g.UP ? 1 : 0, ru = ptid - pz * g.UP;
    const int row = mdiv(ru, g.magic_pw2), pr = ru - row * g.pw2;
    const int gh = h0 + row - g.ph, gw = w0 + 2 * pr - g.pw;
    const bool in_hw = ru < g.UP && (unsigned)gh < (unsigned)g.x.H && (unsigned)gw < (unsigned)g.x.W;
    const unsigned sp_hw = (unsigned)gh * (unsigned)g.x.sH + (unsigned)gw * (unsigned)(XM == 2 ? 2 : 1);
    const int lpos_hw = row * g.PW + 2 * pr;
    auto issue = [&](int T, XT (&pvs)[kCK], unsigned& msk) {                // loads of the planes committed at T
      const Bf3WsPlanes w = bf3_ws_planes_at<KD>(T, nch);
      const int c = pz ? w.c1 : w.c0, p = pz ? w.p1 : w.p0;
      const int gd = d0 + p - g.pd;
      const bool ld = pz < w.n && in_hw && (unsigned)gd < (unsigned)g.x.D && g.dbg != 5;
      msk = ld ? 1u : 0u;
      const int c0 = (cbeg + c) * kCK;
      const unsigned sp = (unsigned)gd * (unsigned)g.x.sD + sp_hw;
      // the chunk's eight channel offsets: two 16-byte LDS reads (one address for the whole plane)
      const u32x4 co0 = *reinterpret_cast<const u32x4*>(choff + c0), co1 = *reinterpret_cast<const u32x4*>(choff + c0 + 4);
      unsigned off[kCK];
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl)
        off[cl] = (ld && c0 + cl < g.x.C) ? ((cl < 4 ? co0[cl & 3] : co1[cl & 3]) + sp) * 4u : 0x80000000u;
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl) XLoad<XM>::load(pvs[cl], xrs, off[cl]);
    };
    auto commit = [&](int T, XT (&pvs)[kCK], unsigned msk) {
      const Bf3WsPlanes w = bf3_ws_planes_at<KD>(T, nch);
      if (pz < w.n && ru < g.UP) {
        const int c = pz ? w.c1 : w.c0, p = pz ? w.p1 : w.p0;
        const int c0 = (cbeg + c) * kCK;
        float v0[kCK], v1[kCK];
#pragma unroll
        for (int cl = 0; cl < kCK; ++cl) { v0[cl] = XLoad<XM>::e0(pvs[cl]); v1[cl] = XLoad<XM>::e1(pvs[cl]); }
        if (g.tr.scale && msk) {                                            // zero padding stays zero
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(tscale + c0), s1 = *reinterpret_cast<const f32x4*>(tscale + c0 + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(tshift + c0), h1 = *reinterpret_cast<const f32x4*>(tshift + c0 + 4);
#pragma unroll
          for (int cl = 0; cl < kCK; ++cl) {
            const float sc = cl < 4 ? s0[cl & 3] : s1[cl & 3], sh = cl < 4 ? h0[cl & 3] : h1[cl & 3];
            float a = v0[cl], e = v1[cl];
            if (g.tr.pre_relu) { a = fmaxf(a, 0.f); e = fmaxf(e, 0.f); }
            a = a * sc + sh; e = e * sc + sh;
            if (g.tr.post_relu) { a = fmaxf(a, 0.f); e = fmaxf(e, 0.f); }
            v0[cl] = a; v1[cl] = e;
          }
        }
        bf16x8 h0v, l0v, h1v, l1v;
        split8(v0, h0v, l0v);
        split8(v1, h1v, l1v);
        const int lpos = ((PD * c + p) & (kWsR - 1)) * g.PHW + lpos_hw;
        Ahi[lpos] = h0v; Ahi[lpos + 1] = h1v;
        Alo[lpos] = l0v; Alo[lpos + 1] = l1v;
      }
    };
    // iteration t: issue the loads of the planes committed at t + 2 into set (t + 2) % 3, then commit the planes of
    // t from set t % 3 once only the two younger sets are still outstanding (every set is always issued in full --
    // lanes without a unit use the out-of-range offset -- so the count is a constant)
    const bool pstamp = g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && ptid == 0;
    auto pmark = [&](int t, int i) { if (pstamp && t < 24) g.stamps[t * 8 + i] = (long long)__builtin_amdgcn_s_memtime(); };
    auto stage = [&](int t, auto S_) {
      constexpr int S = decltype(S_)::value;                   // = t mod 3
      pmark(t, 4);
      issue(t + 2, pv[(S + 2) % 3], inm[(S + 2) % 3]);
      pmark(t, 5);
      wait_loads1d_n<2 * kCK>(pv[S]);
      pmark(t, 6);
      if (g.dbg != 4) commit(t, pv[S], inm[S]);
      pmark(t, 7);
    };
    issue(0, pv[0], inm[0]);                                   // the pipeline is two iterations deep before the loop
    issue(1, pv[1], inm[1]);
    // three iterations per trip, straight-line: registers with loads in flight must never meet a control-flow merge
    // (the compiler would be free to copy them there, before the data has landed)
    for (int t = 0; t < niter; t += 3) {
      __syncthreads();
      stage(t, std::integral_constant<int, 0>());
      __syncthreads();
      stage(t + 1, std::integral_constant<int, 1>());
      __syncthreads();
      stage(t + 2, std::integral_constant<int, 2>());
    }
    return;
  }

  // -------------------------------------------------- consumers --------------------------------------------------
  int toff[NG];
  int pa[kMSUB];
  int sd_w = 0;
  f32x4 acc[kMSUB][NSUB];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    const int t = gq * 4 + kk < g.KHW ? gq * 4 + kk : 0;       // slots past the window carry zero weights
    const int zh = mdiv(t, g.magic_kw), zw = t - zh * g.kw;
    toff[gq] = zh * g.PW + zw;
  }
  {
    const int ri = i16 / g.mw, rj = i16 - ri * g.mw;
#pragma unroll
    for (int ms = 0; ms < kMSUB; ++ms) {
      int s_ = wave * kMSUB + ms;
      const int sw = s_ % g.nsw; s_ /= g.nsw;
      const int sh = s_ % g.nsh;
      sd_w = s_ / g.nsh;                                       // (nsw * nsh is a multiple of kMSUB: the same for all ms)
      pa[ms] = (sh * g.mh + ri) * g.PW + sw * g.mw + rj + g.lead;
    }
  }
#pragma unroll
  for (int ms = 0; ms < kMSUB; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (inline asm: hipcc does not know these loads write LDS, so it neither drains them before the ds_reads of the
  // current step -- which read the OTHER slab buffer -- nor before the barrier; the wait is the explicit vmcnt(0) at
  // the end of the step.  M0 = LDS byte address of the 1 KiB piece; lane l lands at +16*l.)
  const crn_rsrc wsrs = make_rsrc(reinterpret_cast<const float*>(g.wslab));
  const unsigned lds_bhi = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Bhi);
  const unsigned lds_blo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Blo);
  auto slab_dma = [&](int n) {                                 // the slab of step n -> slab buffer n & 1
    const int c = n / KD, zd = n - c * KD;
    const unsigned sbase = (unsigned)((cbeg + c) * g.kd + zd) * (unsigned)(NG * 4);
    const unsigned bufo = (unsigned)((n & 1) * kSlab * 16);
#pragma unroll
    for (int j = 0; j < (kSlabI + kWsConsumers - 1) / kWsConsumers; ++j) {
      const int piece = wave + j * kWsConsumers;               // wave-uniform
      if (piece < kSlabI) {
        const int it = piece * 64 + lane;
        const int nn = it & (NB - 1), tp = it / NB;
        const unsigned off = n0 + nn < g.Npad ? ((sbase + (unsigned)tp) * (unsigned)g.Npad + (unsigned)(n0 + nn)) * 32u : 0x80000000u;
        const unsigned mh = __builtin_amdgcn_readfirstlane(lds_bhi + bufo + (unsigned)piece * 1024u);
        const unsigned ml = __builtin_amdgcn_readfirstlane(lds_blo + bufo + (unsigned)piece * 1024u);
        const unsigned off_lo = off + 16u;                     // (bit 31 survives: out-of-range lanes stay out of range)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
                     "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, 0 offen lds"
                     :: "s"(mh), "s"(ml), "v"(off), "v"(off_lo), "s"(wsrs) : "memory");   // (m0 is not an allocatable register: nothing of the compiler's lives in it here)
      }
    }
  };
  const bool cstamp = g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
  auto cmark = [&](int t, int i) { if (cstamp && t < 24) g.stamps[t * 8 + i] = (long long)__builtin_amdgcn_s_memtime(); };
  for (int t = 0; t < niter; ++t) {
    __syncthreads();             // step t-3's planes and slab are in LDS; nobody reads the step before it any more
    const int n = t - 3;
    cmark(t, 0);
    if (n + 1 >= 0 && n + 1 < nsteps) slab_dma(n + 1);
    cmark(t, 1);
    if (n >= 0 && n < nsteps) {
      const int c = n / KD, zd = n - c * KD;
      const int zr = __builtin_amdgcn_readfirstlane(zrtab[c]);
      if (zd >= (zr & 255) && zd < (zr >> 8) && g.dbg != 1) {  // (outside: only structural zeros)
        const int abase = ((PD * c + zd + sd_w) & (kWsR - 1)) * g.PHW;
        const bf16x8* bh0 = Bhi + (n & 1) * kSlab;
        const bf16x8* bl0 = Blo + (n & 1) * kSlab;
        // software pipeline over "units" u = (tap group gq, half of the wave's sub-tiles): the A fragments of unit
        // u + 1 (and, at a group boundary, the B fragments of the next group) are read before the MFMAs of unit u
        // issue.  NSUB 1: a unit is a whole group (4 sub-tiles, 12 MFMAs); NSUB 2: half a group (2 sub-tiles, 12 MFMAs),
        // which keeps the double-buffered fragments at 64 registers.
        constexpr int MH = NSUB == 1 ? kMSUB : kMSUB / 2, HPG = kMSUB / MH, NU = NG * HPG;
        bf16x8 bh[2][NSUB], bl[2][NSUB], ah[2][MH], al[2][MH];
        auto fragsB = [&](int gq, int q) {
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            bh[q][ns] = bh0[(gq * 4 + kk) * NB + ns * 16 + i16];
            bl[q][ns] = bl0[(gq * 4 + kk) * NB + ns * 16 + i16];
          }
        };
        auto fragsA = [&](int u, int q) {
          const int gq = u / HPG, hf = u - gq * HPG;
          const int off = toff[gq] + abase;
#pragma unroll
          for (int m = 0; m < MH; ++m) { ah[q][m] = Ahi[pa[hf * MH + m] + off]; al[q][m] = Alo[pa[hf * MH + m] + off]; }
        };
        fragsB(0, 0);
        fragsA(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int gq = u / HPG, hf = u - gq * HPG, q = u & 1, qb = gq & 1;
          if (u + 1 < NU) {
            if (hf == HPG - 1) fragsB(gq + 1, qb ^ 1);
            fragsA(u + 1, q ^ 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
              mfma3(acc[hf * MH + m][ns], ah[q][m], al[q][m], bh[qb][ns], bl[qb][ns]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    cmark(t, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next step's slab has landed
    cmark(t, 3);
  }

  // epilogue: D row = kk*4 + r = position (kk*4 + r) of the mh x mw sub-tile, col = i16 = channel
  float* yb = g.y.base + (int64_t)(g.mode >= 3 ? split * g.x.B + b : b) * g.y.sB;
  float bn_s1[NSUB], bn_s2[NSUB], bn_mu[NSUB], bn_rstd[NSUB];
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    bn_s1[ns] = bn_s2[ns] = 0.f;
    const int n = n0 + ns * 16 + i16;
    const bool on = g.bn_x && n < g.y.C;
    bn_mu[ns] = on ? g.bn_saved[n] : 0.f; bn_rstd[ns] = on ? g.bn_saved[g.y.C + n] : 0.f;
  }
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    const int n = n0 + ns * 16 + i16;
    if (n >= g.y.C) continue;
    const int64_t co = view_chan(g.y, n);
    const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
#pragma unroll
    for (int ms = 0; ms < kMSUB; ++ms) {
      int s_ = wave * kMSUB + ms;
      const int sw = s_ % g.nsw; s_ /= g.nsw;
      const int sh = s_ % g.nsh, sd = s_ / g.nsh;
      const int p0 = kk * 4;
      const int od = d0 + sd, oh = h0 + sh * g.mh + p0 / g.mw, ow = w0 + sw * g.mw + p0 % g.mw;
      if (od >= g.y.D || oh >= g.y.H || ow >= g.y.W) continue;
      float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
      if (g.vec_store) {
        f32x4 v = acc[ms][ns] + bsv;
        if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
        if (g.bn_x) {                            // (fused only with vector stores: dense y, dense x of the same shape)
          const f32x4 xv = *reinterpret_cast<const f32x4*>(g.bn_x + (int64_t)b * g.bn_sB + (int64_t)n * g.bn_S +
                                                            ((int64_t)od * g.y.H + oh) * g.y.W + ow);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xr = g.bn_pre_relu ? fmaxf(xv[r], 0.f) : xv[r];
            bn_s1[ns] += v[r];
            bn_s2[ns] += v[r] * ((xr - bn_mu[ns]) * bn_rstd[ns]);
          }
        } else if (g.bn_ws) {                    // forward statistics of the norm BEHIND this conv: sum(y), sum(y^2) (of max(y, 0))
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yr = g.bn_pre_relu ? fmaxf(v[r], 0.f) : v[r];
            bn_s1[ns] += yr;
            bn_s2[ns] += yr * yr;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ow + r < g.y.W) {
            float* d = dst + (int64_t)r * g.y.sW;
            const float v = acc[ms][ns][r] + bsv;
            *d = g.mode == 1 ? *d + v : v;
          }
        }
      }
    }
  }
  if (g.bn_ws) bn_bwd_sums_store<NSUB>(g, smem, bn_s1, bn_s2, wave, kk, i16, n0, tid);
}

// MEASURED AND REMOVED: a "sliding" variant of the kernel above (K = 4 (zd, zw) tap pairs x 8 channels, the zh taps
// in time, so that the KH + 3 patch rows of a wave's four H-consecutive sub-tiles are read once per group: 1.9x
// fewer LDS bytes per MFMA).  Bit-identical results, 0-10 % SLOWER on every decoder layer: the MFMA phase of a
// staging step already runs the matrix pipe at full rate (2 waves x 84 MFMAs in 2900 cycles, CRN_BF3_STAMPS);
// what the kernel loses is the ~3400 cycles per step in which a workgroup issues loads, commits and sits in
// barriers while only the other resident workgroup can feed the pipe.
// Also measured without effect on stage_6.c1 forward (447 us in tools/bench_conv.py): tap offsets as LDS immediates
// (compile-time row pitch, a 5x5 slot order whose groups differ by a per-lane constant; the 70 address adds per
// plane go away, the time does not), MFMAs ordered so that consecutive ones use different accumulators, and a start
// delay for the second resident workgroup (s_sleep of 1-6 k clocks keyed on HW_REG_LDS_ALLOC.lds_base).  Ablations:
// 151 us without the MFMA loop, 418 us without LDS commits, 434 us without global loads after the first step, and
// the MFMA loop with its LDS reads removed still leaves 413 us of a (slower, 525 us) instrumented build: the two
// resident workgroups do not overlap one's staging with the other's MFMAs, T ~ T_mfma + T_staging.  The next
// thing to try is one 16-wave workgroup whose two 8-wave halves alternate roles under block barriers (shared patch,
// window planes split by parity, one LDS reduction at the end).

// ------------------------------------ weight gradient ----------------------------------------------------
// dw[(c*T + t)*Npad + n] += sum over (b, position) of T(x)[b, c, position + t - pad] * dy[b, n, position]
// (the operation of crn_conv_wgrad) on the same split-bf16 MFMA.
//  * GEMM view per window tap: D_t[c][n] = X_t^T . dY, K = output positions.  One MFMA covers 32 positions
//    (two H rows x 16 W) and M = 16 rows = 2 taps x 8 input channels; a workgroup owns 8 input channels
//    (blockIdx.x), NSUB*16 output channels (blockIdx.y) and a slice of the position tiles (blockIdx.z); its 8
//    waves split the tap pairs, so every wave keeps its own (tap pair, n block) accumulators for the whole
//    slice and the partial sums are added to dw with fire-and-forget atomics at the end.
//  * Both operands have K = positions as the slow index of the LDS images ([position][8 channels] for the
//    input patch -- the same image as the forward engine -- and [position][NB] for the dy tile), which is the
//    wrong way round for an MFMA operand register (8 consecutive K per lane): ds_read_b64_tr_b16 transposes
//    4x4 blocks on the way out of LDS, and because every lane supplies its own address a window tap is
//    still just an address offset (whole positions, so the reads stay 8-byte aligned).
constexpr int kWgTile = 512;          // 4 x 8 x 16 positions = 16 K-blocks of 32

struct Bf3WgGeom {
  crnView x, dy;
  crnInTransform tr;
  float* dw;
  int Npad, ncols;
  int kd, kh, kw, pd, ph, pw, T, KHW;
  int PD, PH, PW, PHW, NP, pw2, nunits, lead;
  int tilesD, tilesH, tilesW, ntiles, tiles_per_split;
  unsigned magic_pw2, magic_PH, magic_kw, magic_KHW;
  int dbg;
  int inherit;             // tiles inherit the kd - 1 patch planes they share with the tile below (CRN_BF3_WG_INHERIT=0: off)
  // tap boxes of the output columns (crnTapBoxes, transposed convolutions): a workgroup whose columns share a (d, h) tap range
  // deals only the window rows inside it to its waves (boxskip; kw == 4: two tap pairs per row)
  int boxskip, n_groups;
  signed char n_box[8][6];
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// TWT: tile width: 16 (tile 4 x 8 x 16, a K block of 32 positions = 2 H rows x 16 W) or 8 (tile 8 x 8 x 8 -- the 8^3
// maps of decoder stage 3 --, a K block = 4 H rows x 8 W)
template <int NSUB, int DM, int TPW, int TWT, bool PIPE>
__global__ __launch_bounds__(kThreads) void conv_bf3_wgrad_kernel(Bf3WgGeom g) {
  crn_kernarg_touch(g);
  constexpr int TDT = TWT == 16 ? 4 : 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NB = NSUB * 16;
  constexpr int kHdr = 1024;
  unsigned* choff = reinterpret_cast<unsigned*>(smem);                    // [8] x channel offsets
  float* tscale = reinterpret_cast<float*>(smem + 64);                    // [8]
  float* tshift = reinterpret_cast<float*>(smem + 128);                   // [8]
  unsigned* dchoff = reinterpret_cast<unsigned*>(smem + 256);             // [NB <= 64] dy channel offsets
  char* Xhi = smem + kHdr;
  char* Xlo = Xhi + (size_t)g.NP * 16;
  char* Yhi = Xlo + (size_t)g.NP * 16;                                    // [512][NB] bf16
  char* Ylo = Yhi + (size_t)kWgTile * NB * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  const int c0 = blockIdx.x * kCK, n0 = blockIdx.y * NB;
  const int tbeg = min((int)blockIdx.z * g.tiles_per_split, g.ntiles), tend = min(tbeg + g.tiles_per_split, g.ntiles);
  typedef typename XLoad<DM>::T DT;

  if (tid < kCK) {
    const int c = min(c0 + tid, g.x.C - 1);
    choff[tid] = g.x.chan_off ? (unsigned)g.x.chan_off[c] : (unsigned)c * (unsigned)g.x.sC;
    tscale[tid] = g.tr.scale ? g.tr.scale[c] : 1.f;
    tshift[tid] = g.tr.scale ? g.tr.shift[c] : 0.f;
  }
  if (tid >= 64 && tid < 64 + NB) {
    const int n = min(n0 + tid - 64, g.dy.C - 1);
    dchoff[tid - 64] = g.dy.chan_off ? (unsigned)g.dy.chan_off[n] : (unsigned)n * (unsigned)g.dy.sC;
  }

  // transpose-read source role of this lane: row j of the [4 K][16 M] block of its 16-lane group, column quad q
  const int j = i16 >> 2, q = i16 & 3;
  const int k1 = kk * 8 + j;                                   // K index inside the 32-position block (second read: +4)
  // A (input patch): virtual column quad q = (tap of the pair: q >> 1, channel half: q & 1)
  const int abase = (((k1 / TWT) * g.PW + (k1 % TWT) + g.lead) << 4) + ((q & 1) << 3);
  // Tap pairs of this wave.  Plain: pairs wave * TPW .. + TPW - 1.  boxskip: the window rows (zd, zh) inside the tap box of the
  // workgroup's columns are dealt round-robin -- pair list index wave + 8 ti = (row, half) -- so that every wave multiplies
  // ceil(rows * 2 / 8) pairs instead of TPW (a 7^3 stride-2 transposed convolution: 9 / 12 / 12 / 16 of 16 rows hold real taps)
  int bx_d0 = 0, bx_h0 = 0, bx_nh = 1, bx_np = 0, tpwe = TPW;
  if (!PIPE && g.boxskip) {
    const TapBox tb = box_union(g.n_box, g.n_groups, g.dy.C, n0, min(n0 + NB, g.dy.C) - 1, g.kd, g.kh, g.kw);
    bx_d0 = tb.d0; bx_h0 = tb.h0; bx_nh = max(tb.h1 - tb.h0, 1);
    bx_np = max(tb.d1 - tb.d0, 0) * max(tb.h1 - tb.h0, 0) * 2;
    tpwe = min(TPW, (bx_np + 7) >> 3);
  }
  auto pair_of = [&](int ti) -> int {                          // tap pair index of this wave's slot ti (-1: none)
    if (!(!PIPE && g.boxskip)) return wave * TPW + ti;
    const int idx = wave + 8 * ti;
    if (idx >= bx_np) return -1;
    const int row = idx >> 1, zd = bx_d0 + row / bx_nh, zh = bx_h0 + row % bx_nh;
    return ((zd * g.kh + zh) * g.kw >> 1) + (idx & 1);
  };
  int toffL[TPW];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int pr = pair_of(ti);
    int tp = 2 * max(pr, 0) + (q >> 1);
    if (tp >= g.T) tp = 0;                                     // rows of taps past the window are never stored
    const int zd = mdiv(tp, g.magic_KHW), r = tp - zd * g.KHW;
    const int zh = mdiv(r, g.magic_kw), zw = r - zh * g.kw;
    toffL[ti] = ((zd * g.PH + zh) * g.PW + zw) << 4;
  }
  // B (dy tile, [position][NB] bf16): column quad q of n block ns
  const int ybase = k1 * (NB * 2) + (q << 3);
  const int xlo = g.NP * 16, ylo = kWgTile * NB * 2;

  f32x4 acc[TPW][NSUB];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ti][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  XLoad<1>::T pv[kNUX][kCK];
  DT dv[NSUB][kCK];
  unsigned dco[NSUB][kCK];
  float tsc[kCK], tsh[kCK];
  unsigned inmask = 0;
  int b = 0, d0 = 0, h0 = 0, w0 = 0;
  // Tiles are walked D-first: the next tile of a workgroup is the one ABOVE the current one (same b, h, w), whose patch shares
  // kd - 1 of its PD = TDT + kd - 1 planes with it.  Those planes stay in LDS -- converted, split, transformed -- and move down
  // by TDT planes with an LDS-to-LDS copy; only the TDT new planes are loaded, transformed and split (round 5: the x traffic
  // of stage_6.c1 was 3.75 x the tensor -- every tile re-read its 5^3 halo -- and half of a tile's staging instructions went
  // into planes the workgroup had staged one tile earlier).  ubase / ucount: the unit range a tile stages.
  int ubase = 0;                                               // units of the planes the tile inherits (0: a full patch)
#define ucount (g.nunits - ubase)
  // (the 32-column instances on stride-2 dy views without the software pipeline sit at 250-252 registers: the extra state of
  // the inherited planes would spill their staging registers -- a correctness hazard with loads in flight -- so they stage full patches)
  constexpr bool kInherit = !(NSUB == 2 && DM == 2 && !PIPE);
  auto tile_reuses = [&](int tl) { return kInherit && g.inherit && tl > tbeg && (tl % g.tilesD) != 0; };
  auto tile_origin = [&](int tl) {
    int tile = tl;
    if constexpr (kInherit) {
      const int tdi = tile % g.tilesD; tile /= g.tilesD;
      const int twi = tile % g.tilesW; tile /= g.tilesW;
      const int thi = tile % g.tilesH; tile /= g.tilesH;
      b = tile; d0 = tdi * TDT; h0 = thi * 8; w0 = twi * TWT;
      ubase = tile_reuses(tl) ? (g.PD - TDT) * g.PH * g.pw2 : 0;
    } else {
      const int twi = tile % g.tilesW; tile /= g.tilesW;
      const int thi = tile % g.tilesH; tile /= g.tilesH;
      const int tdi = tile % g.tilesD; tile /= g.tilesD;
      b = tile; d0 = tdi * TDT; h0 = thi * 8; w0 = twi * TWT;
    }
  };
  auto unit_of = [&](int jx, int& pos, unsigned& sp, bool& in) -> bool {
    const int ul = tid + jx * kThreads;                         // index inside the tile's unit range
    int u = ubase + ul;
    asm volatile("" : "+v"(u));
    const int row = mdiv(u, g.magic_pw2), pp = u - row * g.pw2;
    const int pdz = mdiv(row, g.magic_PH), phy = row - pdz * g.PH;
    const int gd = d0 + pdz - g.pd, gh = h0 + phy - g.ph, gw = w0 + 2 * pp - g.pw;
    in = (unsigned)gd < (unsigned)g.x.D && (unsigned)gh < (unsigned)g.x.H && (unsigned)gw < (unsigned)g.x.W;
    sp = (unsigned)gd * (unsigned)g.x.sD + (unsigned)gh * (unsigned)g.x.sH + (unsigned)gw;
    pos = row * g.PW + 2 * pp;
    return ul < ucount;
  };
  // dy unit: (position pair pq of the 256 of a tile, octet o of the NB columns); thread -> (o = jd, pq = tid & 255) ...
  // 512 threads: pair = tid & 255, octet = (tid >> 8) + 2 * jd
  auto dy_unit = [&](int jd, int& pq, int& oct, unsigned& sp, bool& in) {
    pq = tid & 255; oct = (tid >> 8) + 2 * jd;
    const int p = pq * 2;                                       // tile-linear position: (d*8 + h)*TWT + w
    const int w = p % TWT, h = (p / TWT) & 7, d = p / (TWT * 8);
    const int od = d0 + d, oh = h0 + h, ow = w0 + w;
    in = od < g.dy.D && oh < g.dy.H && ow < g.dy.W;
    sp = (unsigned)od * (unsigned)g.dy.sD + (unsigned)oh * (unsigned)g.dy.sH + (unsigned)ow * (unsigned)(DM == 2 ? 2 : 1);
  };
  auto stage_issue = [&]() {
    const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
    const crn_rsrc drs = make_rsrc(g.dy.base + (int64_t)b * g.dy.sB);
    inmask = 0;
#pragma unroll
    for (int jx = 0; jx < kNUX; ++jx) {
      // (no branch around these loads -- registers with loads in flight must not meet a control-flow merge: the units a tile
      // does not stage get the out-of-range offset, which costs an instruction slot and no traffic)
      int pos; unsigned sp; bool in;
      const bool ld = unit_of(jx, pos, sp, in) && in;
      if (ld) inmask |= 1u << jx;
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl) {
        // (x is a plain view here -- even_view, checked on the host: the channel offset is workgroup-uniform scalar arithmetic,
        // not eight registers filled from the LDS table)
        const unsigned xc = (unsigned)min(c0 + cl, g.x.C - 1) * (unsigned)g.x.sC;
        const unsigned off = (ld && c0 + cl < g.x.C) ? (xc + sp) * 4u : 0x80000000u;
        XLoad<1>::load(pv[jx][cl], xrs, off);
      }
    }
#pragma unroll
    for (int jd = 0; jd < NSUB; ++jd) {
      int pq, oct; unsigned sp; bool in;
      dy_unit(jd, pq, oct, sp, in);
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl) {
        const int n = oct * 8 + cl;
        unsigned dc;
        if constexpr (DM == 1) dc = (unsigned)min(n0 + n, g.dy.C - 1) * (unsigned)g.dy.sC;   // (plain dy view: no table, 16 registers less)
        else dc = dco[jd][cl];
        const unsigned off = (in && n0 + n < g.ncols && n0 + n < g.dy.C) ? (dc + sp) * 4u : 0x80000000u;
        XLoad<DM>::load(dv[jd][cl], drs, off);
      }
    }
  };
  auto stage_commit = [&]() {
    if (ubase) {
      // inherited planes: [TDT, PD) of the previous patch become [0, PD - TDT) of this one (disjoint ranges: kd - 1 <= TDT)
      const int n16 = (g.PD - TDT) * g.PHW, src = TDT * g.PHW * 16;
      for (int i = tid; i < n16; i += kThreads) {
        const bf16x8 hv = *reinterpret_cast<const bf16x8*>(Xhi + src + (size_t)i * 16);
        const bf16x8 lv = *reinterpret_cast<const bf16x8*>(Xlo + src + (size_t)i * 16);
        *reinterpret_cast<bf16x8*>(Xhi + (size_t)i * 16) = hv;
        *reinterpret_cast<bf16x8*>(Xlo + (size_t)i * 16) = lv;
      }
      __syncthreads();                                           // the new planes overwrite what was just copied from
    }
#pragma unroll
    for (int jx = 0; jx < kNUX; ++jx) {
      if (jx * kThreads < ucount) {
        int pos; unsigned sp; bool in;
        if (unit_of(jx, pos, sp, in)) {
          float v0[kCK], v1[kCK];
#pragma unroll
          for (int cl = 0; cl < kCK; ++cl) { v0[cl] = pv[jx][cl][0]; v1[cl] = pv[jx][cl][1]; }
          if (g.tr.scale && ((inmask >> jx) & 1u)) {
#pragma unroll
            for (int cl = 0; cl < kCK; ++cl) {
              if (c0 + cl < g.x.C) {
                const float sc = tsc[cl], sh = tsh[cl];
                float a = v0[cl], c = v1[cl];
                if (g.tr.pre_relu) { a = fmaxf(a, 0.f); c = fmaxf(c, 0.f); }
                a = a * sc + sh; c = c * sc + sh;
                if (g.tr.post_relu) { a = fmaxf(a, 0.f); c = fmaxf(c, 0.f); }
                v0[cl] = a; v1[cl] = c;
              }
            }
          }
          bf16x8 h0v, l0v, h1v, l1v;
          split8(v0, h0v, l0v);
          split8(v1, h1v, l1v);
          *reinterpret_cast<bf16x8*>(Xhi + (size_t)pos * 16) = h0v; *reinterpret_cast<bf16x8*>(Xhi + (size_t)pos * 16 + 16) = h1v;
          *reinterpret_cast<bf16x8*>(Xlo + (size_t)pos * 16) = l0v; *reinterpret_cast<bf16x8*>(Xlo + (size_t)pos * 16 + 16) = l1v;
        }
      }
    }
#pragma unroll
    for (int jd = 0; jd < NSUB; ++jd) {
      int pq, oct; unsigned sp; bool in;
      dy_unit(jd, pq, oct, sp, in);
      float v0[kCK], v1[kCK];
#pragma unroll
      for (int cl = 0; cl < kCK; ++cl) { v0[cl] = XLoad<DM>::e0(dv[jd][cl]); v1[cl] = XLoad<DM>::e1(dv[jd][cl]); }
      bf16x8 h0v, l0v, h1v, l1v;
      split8(v0, h0v, l0v);
      split8(v1, h1v, l1v);
      const size_t o = ((size_t)(pq * 2) * NB + oct * 8) * 2;
      *reinterpret_cast<bf16x8*>(Yhi + o) = h0v; *reinterpret_cast<bf16x8*>(Yhi + o + NB * 2) = h1v;
      *reinterpret_cast<bf16x8*>(Ylo + o) = l0v; *reinterpret_cast<bf16x8*>(Ylo + o + NB * 2) = l1v;
    }
  };
  auto trd = [&](const char* p) -> bf16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  };
  auto cat = [](bf16x4 a, bf16x4 c) -> bf16x8 { return __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7); };

  __syncthreads();
  // the workgroup's 8 input channels and this thread's dy columns never change: their offsets and the BatchRenorm
  // scale / shift live in registers (read from LDS per load, every staged load waited for an LDS round trip first)
#pragma unroll
  for (int cl = 0; cl < kCK; ++cl) {        // (uniform addresses: scalar loads into SGPRs, not 16 registers filled from the LDS tables)
    const int c = min(c0 + cl, g.x.C - 1);
    tsc[cl] = g.tr.scale ? g.tr.scale[c] : 1.f;
    tsh[cl] = g.tr.scale ? g.tr.shift[c] : 0.f;
  }
#pragma unroll
  for (int jd = 0; jd < NSUB; ++jd)
#pragma unroll
    for (int cl = 0; cl < kCK; ++cl) dco[jd][cl] = DM == 1 ? 0u : dchoff[((tid >> 8) + 2 * jd) * 8 + cl];
  if (tbeg < tend) { tile_origin(tbeg); stage_issue(); }
  for (int tl = tbeg; tl < tend; ++tl) {
    wait_loads2d(pv);
    wait_loads2d(dv);
    __syncthreads();
    stage_commit();
    __syncthreads();
    if (tl + 1 < tend) { tile_origin(tl + 1); stage_issue(); }
    if (g.dbg == 1) continue;
    if constexpr (!PIPE) {
      // the compiler's own schedule of the reads (measured faster for the 4^3 windows, TPW 4)
      for (int kb = 0; kb < 16; ++kb) {
        const int kbx = TWT == 16 ? (((kb >> 2) * g.PH + (kb & 3) * 2) * g.PW) << 4
                                  : (((kb >> 1) * g.PH + (kb & 1) * 4) * g.PW) << 4;
        const char* yb = Yhi + ybase + kb * (32 * NB * 2);
        bf16x8 bh[NSUB], bl[NSUB];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          bh[ns] = cat(trd(yb + ns * 32), trd(yb + ns * 32 + 4 * NB * 2));
          bl[ns] = cat(trd(yb + ylo + ns * 32), trd(yb + ylo + ns * 32 + 4 * NB * 2));
        }
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
          if (ti >= tpwe) continue;                            // (workgroup-uniform: boxskip)
          const char* xa = Xhi + abase + toffL[ti] + kbx;
          const bf16x8 ah = cat(trd(xa), trd(xa + 64));
          const bf16x8 al = cat(trd(xa + xlo), trd(xa + xlo + 64));
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            mfma3(acc[ti][ns], ah, al, bh[ns], bl[ns]);
          }
        }
      }
      continue;
    }
    // Software-pipelined over (K block kb, tap pair ti): the transposing reads of the NEXT unit's A fragments (and, at
    // the last tap pair of a K block, of the next block's B fragments) are issued before the MFMAs of the current one.
    // The workgroup is alone on its CU (95 KB of LDS), two waves per SIMD: with the compiler's own schedule (read,
    // wait lgkmcnt(0), multiply) every MFMA triple paid an LDS round trip (MfmaUtil 40 %, profiles/r02_conv_mfma_pmc_bf16x3_wgrad.txt).
    auto kbx_of = [&](int kb) {
      return TWT == 16 ? (((kb >> 2) * g.PH + (kb & 3) * 2) * g.PW) << 4      // K block kb = (d, H row pair)
                       : (((kb >> 1) * g.PH + (kb & 1) * 4) * g.PW) << 4;     //            = (d, H row quad)
    };
    bf16x8 bh[2][NSUB], bl[2][NSUB], ah[2], al[2];
    auto ldB = [&](int kb, int qb) {
      const char* yb = Yhi + ybase + kb * (32 * NB * 2);
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        bh[qb][ns] = cat(trd(yb + ns * 32), trd(yb + ns * 32 + 4 * NB * 2));
        bl[qb][ns] = cat(trd(yb + ylo + ns * 32), trd(yb + ylo + ns * 32 + 4 * NB * 2));
      }
    };
    auto ldA = [&](int kbx, int ti, int q) {
      const char* xa = Xhi + abase + toffL[ti] + kbx;
      ah[q] = cat(trd(xa), trd(xa + 64));
      al[q] = cat(trd(xa + xlo), trd(xa + xlo + 64));
    };
    auto block = [&](int kb, auto QB_) {
      constexpr int qb = decltype(QB_)::value;
      const int kbx = kbx_of(kb);
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int q = ti & 1;                                  // (TPW is even: every K block starts on buffer 0)
        if (ti + 1 < TPW) ldA(kbx, ti + 1, q ^ 1);
        else { const int kn = min(kb + 1, 15); ldB(kn, qb ^ 1); ldA(kbx_of(kn), 0, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          mfma3(acc[ti][ns], ah[q], al[q], bh[qb][ns], bl[qb][ns]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(TPW % 2 == 0, "");
    ldB(0, 0);
    ldA(kbx_of(0), 0, 0);
    for (int kb = 0; kb < 16; kb += 2) {
      block(kb, std::integral_constant<int, 0>());
      block(kb + 1, std::integral_constant<int, 1>());
    }
  }

  // D row m = kk*4 + r = (tap of the pair: m >> 3, channel m & 7), col = i16 = n
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = kk * 4 + r;
      const int pr = pair_of(ti);
      const int tp = 2 * pr + (m >> 3), c = c0 + (m & 7);
      if (pr < 0 || ti >= tpwe || tp >= g.T || c >= g.x.C) continue;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const int n = n0 + ns * 16 + i16;
        if (n < g.ncols) atomicAdd(g.dw + ((int64_t)c * g.T + tp) * g.Npad + n, acc[ti][ns][r]);
      }
    }
}

#undef ucount

template <int NSUB, int DM, int TPW, int TWT>
int launch_bf3_wgrad(const Bf3WgGeom& g, dim3 grid, size_t lds, hipStream_t st) {
  // software-pipelined LDS reads: measured per variant (profiles/r03_wgrad_pipe_ab.txt); CRN_BF3_WG_PIPE = 0 / 1 forces
  static const int force = getenv("CRN_BF3_WG_PIPE") ? atoi(getenv("CRN_BF3_WG_PIPE")) : -1;
  static const int force2 = getenv("CRN_BF3_WG_PIPE2") ? atoi(getenv("CRN_BF3_WG_PIPE2")) : -1;   // ... of the 32-column instances only
  const bool pipe = (NSUB == 2 && force2 >= 0) ? force2 != 0 : force >= 0 ? force != 0 : (NSUB == 1 || (TPW == 8 && TWT == 16));
  auto k = pipe ? conv_bf3_wgrad_kernel<NSUB, DM, TPW, TWT, true> : conv_bf3_wgrad_kernel<NSUB, DM, TPW, TWT, false>;
  if (lds > 65536) CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

unsigned magic20b(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }

template <int NSUB, int XM, int NG, int ZS>
int launch_bf3(const Bf3Geom& g, dim3 grid, size_t lds, hipStream_t st) {
  auto k = g.wslab ? conv_bf3_kernel<NSUB, XM, NG, ZS, true> : conv_bf3_kernel<NSUB, XM, NG, ZS, false>;
  if constexpr (NSUB == 1 && XM == 1 && ZS == 1)         // (instantiated where a model layer needs it: stage_6.c1, 28 channels)
    if (g.half_last && g.PW == 20 && g.mw == 16 && g.mh == 1 && g.nsw == 1 && g.nsh == 8) k = g.wslab ? conv_bf3_half_kernel<NSUB, XM, NG, ZS, true> : conv_bf3_half_kernel<NSUB, XM, NG, ZS, false>;
  if (lds > 65536) CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

template <int NSUB, int XM, int NG>
int launch_bf3_ws(const Bf3Geom& g, dim3 grid, size_t lds, hipStream_t st) {
  auto k = conv_bf3_ws_kernel<NSUB, XM, NG>;
  if (lds > 65536) CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(kWsThreads), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

// The producers' timetable, run on the host: every plane a step reads was written (by an earlier iteration) and still
// holds that chunk's data, and no iteration writes a ring slot that the step multiplying at the same time reads.
template <int KD>
bool bf3_ws_timetable_ok(int nch) {
  constexpr int PD = KD + 3;
  int owner_c[kWsR], owner_p[kWsR];
  for (int i = 0; i < kWsR; ++i) owner_c[i] = owner_p[i] = -1;
  const int nsteps = nch * KD;
  for (int t = 0; t <= nsteps + 2; ++t) {
    const int n = t - 3;
    bool reads[kWsR] = {};
    if (n >= 0) {
      const int c = n / KD, zd = n % KD;
      for (int sd = 0; sd < 4; ++sd) {
        const int slot = (PD * c + zd + sd) & (kWsR - 1);
        if (owner_c[slot] != c || owner_p[slot] != zd + sd) return false;
        reads[slot] = true;
      }
    }
    const Bf3WsPlanes w = bf3_ws_planes_at<KD>(t, nch);
    for (int i = 0; i < w.n; ++i) {
      const int wc = i ? w.c1 : w.c0, wp = i ? w.p1 : w.p0;
      const int slot = (PD * wc + wp) & (kWsR - 1);
      if (reads[slot] || wp >= PD) return false;
      owner_c[slot] = wc; owner_p[slot] = wp;
    }
  }
  return true;
}

bool even_view(const crnView& v) {       // 8-byte staging of position pairs on a unit-stride view
  return v.chan_off == nullptr && v.sW == 1 && (v.W & 1) == 0 && (v.sH & 1) == 0 && (v.sD & 1) == 0 && (v.sC & 1) == 0 &&
         (v.sB & 1) == 0 && (((uintptr_t)v.base) & 7) == 0;
}

constexpr size_t kLdsMax = 160 * 1024 - 512;
long long* g_bf3_stamps = nullptr;

}  // namespace

// tuning aid (CRN_BF3_STAMPS=1): shader-clock stamps of workgroup 0 of the last crn_conv_fwd_bf3 launch
// (24 staging steps x 8: loop top, loads landed, barrier, commit done, barrier, next loads issued, MFMAs issued)
#ifdef CRN_TOOLS      // tools/_build/libcorenet_hip_tools.so only (corenet_amd.build.build_tools)
extern "C" int crn_bf3_debug_stamps(long long* out192) {
  if (!g_bf3_stamps) return CRN_EINVAL;
  CRN_HIP(hipDeviceSynchronize());
  CRN_HIP(hipMemcpy(out192, g_bf3_stamps, 24 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
  return CRN_OK;
}
#endif

// Returns CRN_EINVAL for shapes this engine does not cover (the caller keeps the fp32 engine for those).
namespace {
int conv_fwd_bf3_impl(const crnView* x, const crnInTransform* tr, const float* w, const void* wslab, int Npad,
                      const float* bias, int bias_sB, const crnView* y, int kd, int kh, int kw, int pd, int ph, int pw,
                      int accumulate, const crnTapBoxes* boxes, crnStream stream, crnBnBwdFuse* fuse = nullptr);
}
extern "C" int crn_conv_fwd_bf3(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                                const float* bias, int bias_sB, const crnView* y,
                                int kd, int kh, int kw, int pd, int ph, int pw,
                                int accumulate, const crnTapBoxes* boxes, crnStream stream) {
  if (!w) return CRN_EINVAL;
  return conv_fwd_bf3_impl(x, tr, w, nullptr, Npad, bias, bias_sB, y, kd, kh, kw, pd, ph, pw, accumulate, boxes, stream);
}
extern "C" int crn_conv_fwd_bf3_slabs(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                                      const float* bias, int bias_sB, const crnView* y,
                                      int kd, int kh, int kw, int pd, int ph, int pw,
                                      int accumulate, const crnTapBoxes* boxes, crnStream stream) {
  if (!wslab) return CRN_EINVAL;
  return conv_fwd_bf3_impl(x, tr, nullptr, wslab, Npad, bias, bias_sB, y, kd, kh, kw, pd, ph, pw, accumulate, boxes, stream);
}
extern "C" int crn_conv_fwd_bf3_slabs_bnbwd(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                                            const float* bias, int bias_sB, const crnView* y,
                                            int kd, int kh, int kw, int pd, int ph, int pw,
                                            int accumulate, const crnTapBoxes* boxes, crnBnBwdFuse* fuse, crnStream stream) {
  if (!wslab || !fuse || !fuse->x || !fuse->saved || !fuse->ws) return CRN_EINVAL;
  return conv_fwd_bf3_impl(x, tr, nullptr, wslab, Npad, bias, bias_sB, y, kd, kh, kw, pd, ph, pw, accumulate, boxes, stream, fuse);
}
// ... leaving the partial sums of the BatchRenorm BEHIND the convolution (reconstruction_decoder.py:56-60: conv -> ReLU -> norm):
// sum(y'), sum(y'^2) per channel and workgroup, y' = max(y, 0) with pre_relu, in ws[(n * nparts + i) * 2 + {0, 1}] for
// crn_batch_renorm_finalize -- the norm's statistics pass over y is not needed.  *nparts = 0 when the launch could not produce
// them (split reduction, strided y, ws too small): the caller runs crn_batch_renorm_stats; y is written either way.
extern "C" int crn_conv_fwd_bf3_slabs_stats(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                                            const float* bias, int bias_sB, const crnView* y,
                                            int kd, int kh, int kw, int pd, int ph, int pw,
                                            int accumulate, const crnTapBoxes* boxes, int pre_relu, double* ws, size_t ws_bytes,
                                            int* nparts, crnStream stream) {
  if (!wslab || !ws || !nparts) return CRN_EINVAL;
  crnBnBwdFuse f{};
  f.pre_relu = pre_relu; f.ws = ws; f.ws_bytes = ws_bytes;
  const int rc = conv_fwd_bf3_impl(x, tr, nullptr, wslab, Npad, bias, bias_sB, y, kd, kh, kw, pd, ph, pw, accumulate, boxes, stream, &f);
  *nparts = f.nparts;
  return rc;
}

namespace {
int conv_fwd_bf3_impl(const crnView* x, const crnInTransform* tr, const float* w, const void* wslab, int Npad,
                      const float* bias, int bias_sB, const crnView* y, int kd, int kh, int kw, int pd, int ph, int pw,
                      int accumulate, const crnTapBoxes* boxes, crnStream stream, crnBnBwdFuse* fuse) {
  if (fuse) fuse->nparts = 0;
  if (boxes && (boxes->n_groups < 0 || boxes->n_groups > 8 || boxes->c_groups < 0 || boxes->c_groups > 8)) return CRN_EINVAL;
  if (!x || !y || (!w && !wslab) || Npad <= 0 || (Npad & 15) || x->B != y->B || kd < 1 || kh < 1 || kw < 1) return CRN_EINVAL;
  if (y->C > Npad || x->C > kTabC) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool armed = crn_splitk_take_armed();
  { const int rcf = crn_splitk_flush(st); if (rcf != CRN_OK) return rcf; }
  const int xmode = even_view(*x) ? 1 : ((x->sW == 2 && x->chan_off != nullptr && (x->W & 1) == 0) ? 2 : 0);
  if (!xmode) return CRN_EINVAL;
  Bf3Geom g{};
  g.x = *x; g.y = *y;
  g.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
  g.w = w; g.wslab = wslab; g.bias = bias; g.Npad = Npad; g.bias_sB = bias_sB;
  g.lead = ((pw % 2) + 2) % 2;                       // patch rows start on an even column: 8-byte loads stay aligned
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw + g.lead;
  g.T = kd * kh * kw; g.KHW = kh * kw;
  g.NG = (g.KHW + 3) / 4;
  if ((g.NG != 4 && g.NG != 7) || kd > 8) return CRN_EINVAL;           // instantiated: 4x4 and 5x5 window planes
  // tile: 32 sub-tiles of 16 positions.  W >= 16: sub-tile = 16 W positions, tile 4 x 8 x 16; W == 8: sub-tile =
  // 2 x 8 (two H rows), tile 8 x 8 x 8 (a whole 8^3 sample)
  if (y->W % 16 == 0 && y->H >= 8 && y->D >= 4) { g.mw = 16; g.mh = 1; g.nsw = 1; g.nsh = 8; g.TD = 4; }
  else if (y->W == 8 && y->H % 2 == 0 && y->H >= 8 && y->D >= 8) { g.mw = 8; g.mh = 2; g.nsw = 1; g.nsh = 4; g.TD = 8; }
  else return CRN_EINVAL;
  g.TH = g.nsh * g.mh; g.TW = g.nsw * g.mw;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = (g.lead + g.TW + kw - 1 + 1) & ~1;
  g.PHW = g.PH * g.PW; g.NP = g.PD * g.PHW;
  g.pw2 = g.PW / 2; g.nunits = g.PD * g.PH * g.pw2;
  if (g.nunits > kNUX * kThreads) return CRN_EINVAL;
  g.tilesD = crn_cdiv(y->D, g.TD); g.tilesH = crn_cdiv(y->H, g.TH); g.tilesW = crn_cdiv(y->W, g.TW);
  g.nchunks = crn_cdiv(x->C, kCK);
  g.magic_pw2 = magic20b(g.pw2); g.magic_PH = magic20b(g.PH); g.magic_kw = magic20b(kw);
  static const bool no_boxes = getenv("CRN_NO_BOXES") != nullptr;
  bool have_nbox = false;
  if (boxes && !no_boxes) {
    if (boxes->n_groups > 0 && y->C % boxes->n_groups == 0) { g.n_groups = boxes->n_groups; memcpy(g.n_box, boxes->n_box, sizeof(g.n_box)); have_nbox = true; }
    if (boxes->c_groups > 0 && x->C % boxes->c_groups == 0) { g.c_groups = boxes->c_groups; memcpy(g.c_box, boxes->c_box, sizeof(g.c_box)); }
  }
  const int64_t tiles = (int64_t)g.tilesD * g.tilesH * g.tilesW * y->B;
  const int KD = g.NG == 7 ? 5 : 4;                   // window depth of the instantiated whole-window slabs
  g.ctab = (x->C + 15) & ~15;
  auto lds_of = [&](int nsub, int zs) { return (size_t)(3 * g.ctab * 4) + (size_t)2 * g.NP * 16 + (size_t)2 * zs * g.NG * 4 * nsub * 16 * 16; };
  // N block (NSUB x 16 columns per workgroup), measured per layer with the pre-arranged weight slabs (tools/bf3bench.sh
  // with CRN_BF3_NSUB = 1 / 2 / 4, profiles/r02_bf3_nsub_sweep.txt): wide blocks re-use the staged patch, narrow ones
  // keep two workgroups on a CU and (transposed convolutions) skip more structural zeros.  32 columns is within 10 %
  // of the best everywhere; 64 wins on the 16^3 maps (32 tiles: one big workgroup per tile and split-K over the
  // channels beat four narrow ones: stage_4 -18 ... -33 %) and for few input channels without tap boxes (stage_5.c1
  // data gradient), and loses badly on the 64^3 maps.  (Before the slabs, 16-column blocks were the default for
  // every transposed convolution and for Npad <= 32: stage_5.t1 forward 255 us against 189 us now.)
  int NSUB;
  if (Npad <= 16) NSUB = 1;
  else if (Npad <= 32) NSUB = 2;
  else NSUB = ((g.mw == 16 && tiles <= 64) || (!have_nbox && x->C <= 32)) ? 4 : 2;
  while (NSUB > 1 && lds_of(NSUB, 1) > kLdsMax) NSUB >>= 1;
  if (const char* f = getenv("CRN_BF3_NSUB")) NSUB = atoi(f);
  if (NSUB != 1 && NSUB != 2 && NSUB != 4) return CRN_EINVAL;
  // Whole-window slabs (ZS = kd: one staging step per chunk instead of one per window plane) were measured
  // SLOWER than plane-by-plane slabs (s6c1 fwd 584 vs 464 us, s5t1 fwd 343 vs 270 us): the long commit phase
  // stalls all eight waves at once, while short steps let the two waves of a SIMD overlap.  ZS stays 1.
  // Also measured and dropped: double-buffered weight slabs with ONE barrier per step (the next slab is written
  // right after the barrier into the idle buffer): no gain on any layer, and the extra 14 KiB of LDS costs
  // stage_6.c1 forward its second workgroup per CU (556 vs 455 us).  Ablations of that launch (tools/bf3dbg.sh):
  // 462 us complete, 170 us without the MFMA loop, 438 us without global loads, 420 us without LDS commits --
  // the MFMA + LDS-read phase itself runs at ~60 % of the free-running loop of tools/mfma_bf3_loop.hip.
  // Round 3, again with the pre-arranged slabs (a commit is two LDS writes per item) on the 4^3 windows of the transposed
  // convolutions (us, ZS 1 -> 4, B = 4, alone): data gradients stage_6.t1 153 -> 147, stage_5.t1 168 -> 165, 14 classes
  // 909 -> 841; forwards 154 -> 185, 194 -> 229, 817 -> 996; inside the step ZS = 4 on the data gradients alone: 7.72 vs
  // 7.62 ms (the 24 KiB of extra LDS meet the weight-gradient kernels of the other stream).  ZS stays 1.
  const int ZS = 1;
  (void)KD;
  const size_t lds = lds_of(NSUB, ZS);
  if (lds > kLdsMax) return CRN_EINVAL;
  if ((int64_t)x->C * g.T * Npad >= ((int64_t)1 << 29)) return CRN_EINVAL;
  // split-K over the channel chunks: partial sums to the shared scratch, one reduction launch
  const int64_t blocks = tiles * crn_cdiv(Npad, NSUB * 16);
  int splits = 1;
  if (blocks < 192) splits = (int)std::min<int64_t>(std::min(g.nchunks, 16), crn_cdiv(256, blocks));
  if (const char* f = getenv("CRN_BF3_SPLITS")) splits = std::max(1, std::min(atoi(f), g.nchunks));
  g.chunks_per_split = crn_cdiv(g.nchunks, splits);
  splits = crn_cdiv(g.nchunks, g.chunks_per_split);
  g.mode = accumulate ? 1 : 0;
  const crnView yreal = *y;
  float* scratch = nullptr;
  // the partial sums: added up by a reduction launch -- or, after crn_splitk_defer, by the BatchRenorm launch that reads
  // this (dense) output next (decoder stages 2-4: statistics after c1, the norms' backward after both data gradients)
  auto finish_splits = [&]() -> int {
    const int64_t S = (int64_t)yreal.D * yreal.H * yreal.W;
    if (armed && !accumulate && yreal.chan_off == nullptr && yreal.sW == 1 && yreal.sH == yreal.W &&
        yreal.sD == (int64_t)yreal.H * yreal.W && yreal.sC == S && yreal.sB == (int64_t)yreal.C * S) {
      crn_splitk_set_pending(yreal, scratch, splits, st);
      return CRN_OK;
    }
    return crn_splitk_reduce(yreal, scratch, splits, accumulate, st);
  };
  if (splits > 1) {
    const int64_t S = (int64_t)y->D * y->H * y->W, ytot = (int64_t)y->B * y->C * S;
    scratch = crn_splitk_scratch((size_t)splits * ytot, st);
    if (!scratch) { splits = 1; g.chunks_per_split = g.nchunks; }
    else {
      g.mode = 3;
      g.yreal = *y; g.accumulate_real = accumulate ? 1 : 0;
      static const bool sk_launch = getenv("CRN_SPLITK_FUSED") == nullptr;   // default: separate reduction launch
      if (!sk_launch && (g.counters = crn_splitk_counters((size_t)tiles * crn_cdiv(Npad, NSUB * 16))) != nullptr) g.mode = 4;
      g.y.base = scratch; g.y.chan_off = nullptr;
      g.y.sW = 1; g.y.sH = y->W; g.y.sD = y->H * y->W; g.y.sC = S; g.y.sB = (int64_t)y->C * S;
    }
  }
  const crnView& yo = g.y;
  g.vec_store = (g.mw >= 4 && yo.sW == 1 && (yo.W & 3) == 0 && (yo.sH & 3) == 0 && (yo.sD & 3) == 0 && (yo.sB & 3) == 0 &&
                 (yo.sC & 3) == 0 && (((uintptr_t)yo.base) & 15) == 0 && yo.chan_off == nullptr) ? 1 : 0;
  const int64_t tiles_bn = tiles;
  // two users of this epilogue, each behind its own switch: the BatchRenorm BACKWARD sums of a data gradient
  // (fuse->x = the norm's input; CRN_BN_BWD_FUSE=0) and the forward STATISTICS of the next norm (fuse->x == nullptr:
  // crn_conv_fwd_bf3_slabs_stats; CRN_DEC_STATS_FUSE=0), which knows nothing of x / sB_x / ndsum
  static const bool bn_fuse_off = getenv("CRN_BN_BWD_FUSE") != nullptr && atoi(getenv("CRN_BN_BWD_FUSE")) == 0;
  static const bool stats_fuse_off = getenv("CRN_DEC_STATS_FUSE") != nullptr && atoi(getenv("CRN_DEC_STATS_FUSE")) == 0;
  const bool stats_mode = fuse && fuse->x == nullptr;
  const bool mode_ok = fuse && (stats_mode ? !stats_fuse_off
                                           : (!bn_fuse_off && fuse->ndsum <= kThreads && (fuse->sB_x & 3) == 0 &&
                                              (((uintptr_t)fuse->x) & 15) == 0));
  if (mode_ok && splits == 1 && g.vec_store && y->sW == 1 && tiles_bn <= 0x7fffffff &&
      fuse->ws_bytes >= (size_t)y->C * (size_t)tiles_bn * 2 * sizeof(double) && (((int64_t)y->D * y->H * y->W) & 3) == 0) {
    g.bn_x = fuse->x; g.bn_sB = fuse->sB_x; g.bn_S = (int64_t)y->D * y->H * y->W; g.bn_saved = fuse->saved;
    g.bn_pre_relu = fuse->pre_relu; g.bn_ws = fuse->ws; g.bn_dsum = fuse->dsum; g.bn_ndsum = fuse->dsum ? fuse->ndsum : 0;
    fuse->nparts = (int)tiles_bn;
  }
  g.dbg = getenv("CRN_DBG_MODE") ? atoi(getenv("CRN_DBG_MODE")) : 0;
  static const bool rowskip_off = getenv("CRN_BF3_ROWSKIP") != nullptr && atoi(getenv("CRN_BF3_ROWSKIP")) == 0;
  g.rowskip = (!rowskip_off && kh == 4 && kw == 4 && (g.n_groups > 0 || g.c_groups > 0)) ? 1 : 0;
  static const bool half_off = getenv("CRN_BF3_HALF") != nullptr && atoi(getenv("CRN_BF3_HALF")) == 0;
  g.half_last = (!half_off && x->C % kCK >= 1 && x->C % kCK <= 4 && ZS == 1) ? 1 : 0;
#ifdef CRN_TOOLS
  static const bool want_stamps = getenv("CRN_BF3_STAMPS") != nullptr;
#else
  constexpr bool want_stamps = false;
#endif
  if (want_stamps) {
    if (!g_bf3_stamps) CRN_HIP(hipMalloc(&g_bf3_stamps, 24 * 8 * sizeof(long long)));
    CRN_HIP(hipMemsetAsync(g_bf3_stamps, 0, 24 * 8 * sizeof(long long), st));
    g.stamps = g_bf3_stamps;
  }
  dim3 grid((unsigned)tiles, (unsigned)crn_cdiv(Npad, NSUB * 16), (unsigned)splits);
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_fwd_bf3] x(C%d %dx%dx%d) y(C%d %dx%dx%d) k%dx%dx%d: NSUB %d ZS %d xmode %d tile %dx%dx%d patch "
            "%dx%dx%d units %d NG %d grid %ux%ux%u lds %zu\n", x->C, x->D, x->H, x->W, y->C, y->D, y->H, y->W, kd, kh, kw, NSUB,
            ZS, xmode, g.TD, g.TH, g.TW, g.PD, g.PH, g.PW, g.nunits, g.NG, grid.x, grid.y, grid.z, lds);
  int rc = CRN_EINVAL;
  // the wave-specialised kernel (conv_bf3_ws_kernel): pre-arranged slabs, 4 x 8 x 16 tiles, N blocks of 16 / 32
  // columns, cubic 5^3 / 4^3 windows
  // Where it is used is measured (tools/ws_ab.sh, profiles/r03_ws_ab.txt, B = 4): it wins where a step is long -- N blocks
  // of 32 columns on 5^3 windows (stage_6.c1 data gradient 359 -> 336 us, stage_5.c1 forward 181 -> 161 us) -- and loses
  // on the 16-column blocks (stage_6.c1 forward 426 -> 486 us, stage_6.t1 160 -> 190 us): there a step is 2.7 k cycles of
  // MFMA fed by 2.2 k cycles of LDS reads (one A fragment pair per three MFMAs), and the ~0.9 k cycles of DMA issue, slab
  // wait and barrier per step cost more than two co-resident workgroups of the kernel above lose.  CRN_BF3_WS = 0 / 1:
  // never / wherever it applies.
  static const int ws_force = getenv("CRN_BF3_WS") ? atoi(getenv("CRN_BF3_WS")) : -1;
  const bool ws_on = ws_force >= 0 ? ws_force != 0 : (NSUB == 2 && g.NG == 7);
  if (ws_on && wslab && xmode == 1 && g.mw == 16 && NSUB <= 2 && g.mode != 4 && kd == (g.NG == 7 ? 5 : 4)) {
    Bf3Geom gw = g;
    gw.UP = gw.PH * gw.pw2; gw.magic_UP = magic20b(gw.UP);
    const int NB = NSUB * 16;
    const size_t lds_ws = (size_t)(3 * gw.ctab * 4) + 256 + (size_t)2 * kWsR * gw.PHW * 16 + (size_t)4 * gw.NG * 4 * NB * 16;
    const int nch = std::min(g.chunks_per_split, g.nchunks);
    const bool ok = 2 * gw.UP <= kWsPL && lds_ws <= kLdsMax && nch <= 64 &&
                    (kd == 5 ? bf3_ws_timetable_ok<5>(nch) : bf3_ws_timetable_ok<4>(nch));
    if (ok) {
      if (dbg) fprintf(stderr, "[crn_conv_fwd_bf3] wave-specialised: %d chunks x %d planes, lds %zu\n", nch, kd, lds_ws);
#define CRN_BF3_WS_CASE(N, X, G) if (NSUB == N && xmode == X && gw.NG == G) rc = launch_bf3_ws<N, X, G>(gw, grid, lds_ws, st);
      CRN_BF3_WS_CASE(1, 1, 7) CRN_BF3_WS_CASE(2, 1, 7)
      CRN_BF3_WS_CASE(1, 1, 4) CRN_BF3_WS_CASE(2, 1, 4)
#undef CRN_BF3_WS_CASE
      if (rc == CRN_OK && g.mode == 3) rc = finish_splits();
      return rc;
    }
  }
#define CRN_BF3_CASE(N, X, G, Z) if (NSUB == N && xmode == X && g.NG == G && ZS == Z) rc = launch_bf3<N, X, G, Z>(g, grid, lds, st);
  CRN_BF3_CASE(1, 1, 7, 1) CRN_BF3_CASE(2, 1, 7, 1) CRN_BF3_CASE(4, 1, 7, 1)
  CRN_BF3_CASE(1, 1, 4, 1) CRN_BF3_CASE(2, 1, 4, 1) CRN_BF3_CASE(4, 1, 4, 1)
  CRN_BF3_CASE(1, 2, 4, 1) CRN_BF3_CASE(2, 2, 4, 1) CRN_BF3_CASE(4, 2, 4, 1)
#undef CRN_BF3_CASE
  if (rc == CRN_OK && g.mode == 3) rc = finish_splits();
  return rc;
}
}  // namespace

// Weight gradient on the split-bf16 MFMA engine; same contract as crn_conv_wgrad (dw zeroed by the caller or
// zero_first).  Returns CRN_EINVAL for shapes it does not cover.
extern "C" int crn_conv_wgrad_bf3_boxes(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                                        int kd, int kh, int kw, int pd, int ph, int pw, int zero_first,
                                        const crnTapBoxes* boxes, crnStream stream);
extern "C" int crn_conv_wgrad_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                                  int kd, int kh, int kw, int pd, int ph, int pw, int zero_first, crnStream stream) {
  return crn_conv_wgrad_bf3_boxes(x, tr, dy, dw, Npad, kd, kh, kw, pd, ph, pw, zero_first, nullptr, stream);
}
// ... with the tap boxes of the output columns (transposed convolutions, conv_geometry.convt_fwd): window rows that hold only
// structural zeros for a workgroup's columns are not multiplied (their dw entries stay what they were: zero)
extern "C" int crn_conv_wgrad_bf3_boxes(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                                        int kd, int kh, int kw, int pd, int ph, int pw, int zero_first,
                                        const crnTapBoxes* boxes, crnStream stream) {
  CRN_ENTRY(stream);
  if (boxes && (boxes->n_groups < 0 || boxes->n_groups > 8 || boxes->c_groups < 0 || boxes->c_groups > 8)) return CRN_EINVAL;
  if (!x || !dy || !dw || Npad <= 0 || (Npad & 15) || x->B != dy->B || kd < 1 || kh < 1 || kw < 1) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (!even_view(*x)) return CRN_EINVAL;
  const int dmode = even_view(*dy) ? 1 : ((dy->sW == 2 && dy->chan_off != nullptr && (dy->W & 1) == 0) ? 2 : 0);
  if (!dmode) return CRN_EINVAL;
  const int TWT = (dy->W % 16 == 0) ? 16 : 8, TDT = TWT == 16 ? 4 : 8;
  if (dy->W % TWT || dy->H < 8 || dy->D < TDT) return CRN_EINVAL;
  Bf3WgGeom g{};
  g.x = *x; g.dy = *dy;
  g.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
  g.dw = dw; g.Npad = Npad; g.ncols = std::min(Npad, dy->C);
  g.lead = ((pw % 2) + 2) % 2;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw + g.lead;
  g.T = kd * kh * kw; g.KHW = kh * kw;
  g.PD = TDT + kd - 1; g.PH = 8 + kh - 1; g.PW = (g.lead + TWT + kw - 1 + 1) & ~1;
  g.PHW = g.PH * g.PW; g.NP = g.PD * g.PHW;
  g.pw2 = g.PW / 2; g.nunits = g.PD * g.PH * g.pw2;
  if (g.nunits > kNUX * kThreads) return CRN_EINVAL;
  const int TPW = (((g.T + 1) / 2) + 7) / 8;                  // tap pairs per wave
  if (TPW != 8 && TPW != 4) return CRN_EINVAL;                // instantiated: 5^3 (63 pairs) and 4^3 (32 pairs) windows
  g.tilesD = crn_cdiv(dy->D, TDT); g.tilesH = crn_cdiv(dy->H, 8); g.tilesW = dy->W / TWT;
  g.ntiles = g.tilesD * g.tilesH * g.tilesW * dy->B;
  g.magic_pw2 = magic20b(g.pw2); g.magic_PH = magic20b(g.PH); g.magic_kw = magic20b(kw); g.magic_KHW = magic20b(g.KHW);
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * g.T * Npad * 4, st));
  int NSUB = (Npad % 32 == 0) ? 2 : 1;
  if (const char* f = getenv("CRN_BF3_WG_NSUB")) NSUB = atoi(f);
  if (NSUB != 1 && NSUB != 2) return CRN_EINVAL;
  const int NB = NSUB * 16;
  const size_t lds = 1024 + (size_t)2 * g.NP * 16 + (size_t)2 * kWgTile * NB * 2;
  if (lds > 160 * 1024 - 512) return CRN_EINVAL;
  const int cblocks = crn_cdiv(x->C, kCK), nblocks = crn_cdiv(Npad, NB);
  static const int kBlocks = getenv("CRN_BF3_WG_BLOCKS") ? atoi(getenv("CRN_BF3_WG_BLOCKS")) : 256;
  int splits = std::max(1, std::min(g.ntiles, kBlocks / std::max(1, cblocks * nblocks)));
  if (crn_deterministic()) splits = 1;                       // one workgroup per dw element: one add, no order to vary
  g.tiles_per_split = crn_cdiv(g.ntiles, splits);
  splits = crn_cdiv(g.ntiles, g.tiles_per_split);
  g.dbg = getenv("CRN_DBG_MODE") ? atoi(getenv("CRN_DBG_MODE")) : 0;
  static const int inherit_env = getenv("CRN_BF3_WG_INHERIT") ? atoi(getenv("CRN_BF3_WG_INHERIT")) : 1;
  // the in-LDS plane copy [TDT, PD) -> [0, PD - TDT) runs on all threads at once: only disjoint ranges (kd - 1 <= TDT) may
  // inherit (every model layer; a (7,3,3) window over 4-plane tiles would race -- ADVICE r5), others stage full patches
  g.inherit = (inherit_env && kd - 1 <= TDT) ? 1 : 0;
  static const bool boxskip_off = getenv("CRN_BF3_WG_BOXSKIP") != nullptr && atoi(getenv("CRN_BF3_WG_BOXSKIP")) == 0;
  if (boxes && !boxskip_off && boxes->n_groups > 0 && dy->C % boxes->n_groups == 0 && kw == 4) {
    g.boxskip = 1; g.n_groups = boxes->n_groups; memcpy(g.n_box, boxes->n_box, sizeof(g.n_box));
  }
  dim3 grid((unsigned)cblocks, (unsigned)nblocks, (unsigned)splits);
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_wgrad_bf3] x(C%d %dx%dx%d) dy(C%d %dx%dx%d) k%dx%dx%d: NSUB %d dmode %d TPW %d grid %ux%ux%u "
            "tiles/split %d lds %zu\n", x->C, x->D, x->H, x->W, dy->C, dy->D, dy->H, dy->W, kd, kh, kw, NSUB, dmode, TPW,
            grid.x, grid.y, grid.z, g.tiles_per_split, lds);
#define CRN_BF3WG_CASE(N, D, P, W) if (NSUB == N && dmode == D && TPW == P && TWT == W) return launch_bf3_wgrad<N, D, P, W>(g, grid, lds, st);
  CRN_BF3WG_CASE(1, 1, 8, 16) CRN_BF3WG_CASE(2, 1, 8, 16)
  CRN_BF3WG_CASE(1, 2, 4, 16) CRN_BF3WG_CASE(2, 2, 4, 16)
  CRN_BF3WG_CASE(1, 1, 4, 16) CRN_BF3WG_CASE(2, 1, 4, 16)
  CRN_BF3WG_CASE(1, 1, 8, 8) CRN_BF3WG_CASE(2, 1, 8, 8)
  CRN_BF3WG_CASE(1, 2, 4, 8) CRN_BF3WG_CASE(2, 2, 4, 8)
#undef CRN_BF3WG_CASE
  return CRN_EINVAL;
}
