// Split-bf16 MFMA convolution, ring-buffered and DMA-fed (gfx950 / CDNA4): forward pass and data gradient of the big
// decoder layers (Conv3d k5 / ConvTranspose3d k7 s2 of decoder stages 4-6, reconstruction_decoder.py:72-95).
//
// Same products, same summation order, same weight slabs as conv_bf3_kernel / conv_bf3_ws_kernel (conv_bf3.hip):
// bit-identical results.  What changes is who prepares the operands and when:
//
//  * The ACTIVATIONS arrive pre-split.  crn_bf3_act_image applies the producer's BatchRenorm + ReLU once per tensor and
//    writes the "activation image": [batch][chunk of 8 channels][D][H][W] entries of 8 bf16 (16 bytes), a hi image and a
//    lo image (x = hi + lo).  It is the LDS patch format of the conv kernels, so staging a patch plane is a plain copy:
//    one LDS-DMA instruction (buffer_load_dwordx4 ... lds) per 64 positions, no registers, no VALU, no LDS write
//    instruction, and zero padding is the out-of-range offset of the buffer descriptor.  Round 3's stamps said a staging
//    step of conv_bf3_kernel is ~6 k cycles of which 2.9 k multiply; the rest (load issue, 2 us of load latency, the
//    split in VALU, the LDS commit, two barriers) is what this removes from the conv kernel; the image is written once
//    and read by the forward convolution AND the weight gradient of the layer.
//  * The patch is a RING of R window planes (R = 12; 8 KiB each) filled by four producer waves (one per SIMD) that run
//    AHEAD of the eight consumer waves: plane P may be written as soon as plane P - R is dead, which is 4 - 8 steps
//    before the step that needs it, so the ~2 us a load takes on the busy chip are never waited for.  The producers
//    wait with a COUNTED vmcnt: only the planes of the next step have to have landed at the barrier.
//  * The workgroup is PERSISTENT: it walks a contiguous range of (tile, N block) items, and the ring does not know where
//    a tile ends -- the planes of the next tile are in flight while the last steps of the current one multiply, and the
//    pipeline is filled once per workgroup instead of once per tile (conv_bf3_ws_kernel: 3-5 iterations of fill for the
//    10-20 steps of a tile, with nothing else resident on the CU to cover them).
//  * The consumers are the consumers of conv_bf3_ws_kernel: one (chunk, window plane) step per barrier, software-pipelined
//    fragment reads, weight slabs double-buffered by LDS-DMA from the pre-arranged slab images (crn_bf3_operands).
#include "conv_kernels.h"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace crnk;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kCK = 8, kMSUB = 4;
constexpr int kCons = 8, kProd = 4, kRingThreads = (kCons + kProd) * 64;
constexpr int kPL = 256;               // positions per ring slot (a patch plane is PH * PW <= 256 positions)
constexpr int kHdr = 1024;             // z-range table in front of the ring
constexpr size_t kLdsMax = 160 * 1024 - 512;

struct RingGeom {
  const void* img_hi;                  // activation image, hi terms; entry (((b*NCH + ch)*D + d)*H + h)*W + w, 16 bytes
  const void* img_lo;                  // ... lo terms
  int B, C, NCH, D, H, W;              // logical input
  crnView y;
  const float* bias; int bias_sB;
  const void* wslab; int Npad;
  int kd, kh, kw, pd, ph, pw, KHW;
  int PH, PW, PHW;
  int tilesD, tilesH, tilesW, NBK;     // NBK = N blocks per tile
  int nitems, items_per_wg;
  int nch, R;
  int mode;                            // 0 store, 1 accumulate
  int vec_store;
  int n_groups, c_groups;
  signed char n_box[8][6], c_box[8][6];
  unsigned magic_kw, magic_PW;
  int dbg;
  long long* stamps;                   // tuning aid (CRN_RING_STAMPS=1): cycle sums of workgroup 0, consumer wave 0 / producer wave 8
};

// x = hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi): the split of conv_bf3.hip
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// ------------------------------------------------ activation image ------------------------------------------------
struct ImageGeom {
  crnView x;
  crnInTransform tr;
  bf16x8* hi; bf16x8* lo;
  int NCH, S, HW;
};

__global__ __launch_bounds__(256) void bf3_act_image_kernel(ImageGeom g) {
  crn_kernarg_touch(g);
  const int p = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, b = blockIdx.z;
  if (p >= g.S) return;
  const int d = p / g.HW, r = p - d * g.HW, h = r / g.x.W, w = r - h * g.x.W;
  const float* xb = g.x.base + (int64_t)b * g.x.sB + (int64_t)d * g.x.sD + (int64_t)h * g.x.sH + (int64_t)w * g.x.sW;
  float v[kCK];
#pragma unroll
  for (int cl = 0; cl < kCK; ++cl) {
    const int c = ch * kCK + cl;
    v[cl] = c < g.x.C ? xb[view_chan(g.x, c)] : 0.f;
  }
  if (g.tr.scale) {
#pragma unroll
    for (int cl = 0; cl < kCK; ++cl) {
      const int c = ch * kCK + cl;
      if (c < g.x.C) {
        float a = v[cl];
        if (g.tr.pre_relu) a = fmaxf(a, 0.f);
        a = a * g.tr.scale[c] + g.tr.shift[c];
        if (g.tr.post_relu) a = fmaxf(a, 0.f);
        v[cl] = a;
      }
    }
  }
  bf16x8 hv, lv;
  split8(v, hv, lv);
  const int64_t e = ((int64_t)b * g.NCH + ch) * g.S + p;
  g.hi[e] = hv;
  g.lo[e] = lv;
}

// ------------------------------------------------ the convolution ------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform, any value >= 0)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  if (n >= 16) { if (n >= 24) wait_vmcnt<24>(); else if (n >= 20) wait_vmcnt<20>(); else if (n >= 18) wait_vmcnt<18>(); else wait_vmcnt<16>(); }
  else if (n >= 8) { if (n >= 14) wait_vmcnt<14>(); else if (n >= 12) wait_vmcnt<12>(); else if (n >= 10) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
  else if (n >= 4) { if (n >= 6) wait_vmcnt<6>(); else wait_vmcnt<4>(); }
  else if (n >= 2) wait_vmcnt<2>();
  else wait_vmcnt<0>();
}

template <int NSUB, int NG, bool kSlide, bool kFlags>
__global__ __launch_bounds__(kRingThreads) void conv_bf3_ring_kernel(RingGeom g) {
  crn_kernarg_touch(g);
  constexpr int NB = NSUB * 16;
  constexpr int KD = NG == 7 ? 5 : 4, PD = KD + 3;             // 5^3 / 4^3 windows on 4 x 8 x 16 tiles
  constexpr int kPW = 16 + KD - 1;                             // patch row pitch (positions): 16 + kw - 1
  constexpr int kSlab = NG * 4 * NB;                           // weight items (hi 16 B + lo 16 B) per slab
  static_assert(kSlab % 64 == 0, "a slab is a whole number of 64-lane DMA instructions");
  constexpr int kSlabI = kSlab / 64;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int* zrtab = reinterpret_cast<int*>(smem);                   // [NBK * nch] z range (z0 | z1 << 8) of a (N block, chunk)
  // kFlags: the waves meet through words in LDS instead of one workgroup barrier per step (below)
  int* f_planes = reinterpret_cast<int*>(smem + kHdr - 128);   // [4] planes landed, per producer wave (its quarter of each plane)
  int* f_slabs = f_planes + 4;                                 // [4] slabs landed, per producer wave (its pieces of each slab)
  int* f_done = f_planes + 8;                                  // [8] steps completed, per consumer wave
  int* f_abort = f_planes + 16;                                // a spin ran out: every wave leaves
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(smem + kHdr);        // [R][kPL]
  bf16x8* Alo = Ahi + g.R * kPL;
  bf16x8* Bhi = Alo + g.R * kPL;                               // [3][kSlab]
  bf16x8* Blo = Bhi + 3 * kSlab;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;
  const int R = g.R, nch = g.nch;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int it_beg = min(wg * g.items_per_wg, g.nitems), it_end = min(it_beg + g.items_per_wg, g.nitems);
  const int nit = it_end - it_beg;
  if (nit <= 0) return;
  const int total_steps = nit * nch * KD, total_planes = nit * nch * PD;

  for (int i = tid; i < g.NBK * nch; i += kRingThreads) {      // window planes that can hold non-zero weights (tap boxes)
    const int nb = i / nch, c = i - nb * nch;
    const TapBox nbox = box_union(g.n_box, g.n_groups, g.y.C, nb * NB, min(nb * NB + NB, g.y.C) - 1, g.kd, g.kh, g.kw);
    const int chi = min(c * kCK + kCK, g.C) - 1;
    const TapBox tb = box_intersect(nbox, box_union(g.c_box, g.c_groups, g.C, c * kCK, chi, g.kd, g.kh, g.kw));
    int z0 = tb.d0, z1 = tb.d1;
    if (tb.h1 <= tb.h0 || tb.w1 <= tb.w0) z1 = z0;
    zrtab[i] = z0 | (z1 << 8);
  }
  if (tid < 32) f_planes[tid] = 0;
  __syncthreads();

  auto decode = [&](int item, int& b, int& d0, int& h0, int& w0, int& nb) {
    nb = item % g.NBK;
    int tile = item / g.NBK;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD;
    b = tile / g.tilesD; d0 = tdi * 4; h0 = thi * 8; w0 = twi * 16;
  };

  if (wave >= kCons) {
    // ------------------------------------------------ producers ------------------------------------------------
    // wave pw owns positions [64 pw, 64 pw + 64) of every plane: one DMA instruction for the hi image, one for the lo
    const int pwv = wave - kCons;
    const int q = pwv * 64 + lane;
    const int prow = mdiv(q, g.magic_PW), pcol = q - prow * g.PW;
    const crn_rsrc rs_hi = make_rsrc(reinterpret_cast<const float*>(g.img_hi));
    const crn_rsrc rs_lo = make_rsrc(reinterpret_cast<const float*>(g.img_lo));
    const unsigned lds_ahi = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Ahi) + pwv * 1024u;
    const unsigned lds_alo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Alo) + pwv * 1024u;
    const unsigned plane_bytes = (unsigned)g.H * (unsigned)g.W * 16u;
    int iP = 0, i_k = 0, i_c = 0, i_p = 0, i_slot = 0;         // the issue pointer: next plane, its item / chunk / plane / slot
    int i_b = 0, i_d0 = 0;
    unsigned voff_hw = 0x80000000u;
    auto set_item = [&](int k) {
      int b, d0, h0, w0, nb;
      decode(it_beg + k, b, d0, h0, w0, nb);
      const int gh = h0 - g.ph + prow, gw = w0 - g.pw + pcol;
      const bool in = q < g.PHW && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
      voff_hw = in ? ((unsigned)gh * (unsigned)g.W + (unsigned)gw) * 16u : 0x80000000u;
      i_b = b; i_d0 = d0;
    };
    auto issue_plane = [&]() {
      const int gd = i_d0 - g.pd + i_p;
      const bool dok = (unsigned)gd < (unsigned)g.D && g.dbg != 5;
      const unsigned soff = (unsigned)((i_b * g.NCH + i_c) * g.D + gd) * plane_bytes;
      const unsigned vo = dok ? voff_hw + soff : 0x80000000u;  // (an out-of-range voff_hw keeps bit 31: soff < 2^31)
      const unsigned mh = __builtin_amdgcn_readfirstlane(lds_ahi + (unsigned)i_slot * (kPL * 16u));
      const unsigned ml = __builtin_amdgcn_readfirstlane(lds_alo + (unsigned)i_slot * (kPL * 16u));
      // (s_nop 4: the descriptor may just have been rebuilt from a spilled SGPR by v_readlane -- 5 wait states before a
      // VMEM instruction reads it, and the hazard recogniser does not look inside inline asm)
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                   "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds"
                   :: "s"(mh), "s"(ml), "v"(vo), "s"(rs_hi), "s"(rs_lo) : "memory");
      ++iP;
      if (++i_slot == R) i_slot = 0;
      if (++i_p == PD) {
        i_p = 0;
        if (++i_c == nch) { i_c = 0; if (++i_k < nit) set_item(i_k); }
      }
    };
    // the weight slabs too: slab s = (item, chunk, window plane) of step s goes to slab buffer s % 3, one iteration before
    // the iteration whose closing barrier publishes it (two iterations before the step multiplies)
    const crn_rsrc wsrs = make_rsrc(reinterpret_cast<const float*>(g.wslab));
    const unsigned lds_bhi = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Bhi);
    const unsigned lds_blo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(Blo);
    int s_k = 0, s_c = 0, s_z = 0, s_nb = 0, s_buf = 0, s_n = 0;      // the slab pointer: next slab, its item / chunk / plane / buffer
    { int b_, d_, h_, w_; decode(it_beg, b_, d_, h_, w_, s_nb); }
    int issued = 0;                                            // DMA instructions of this wave in the current iteration
    auto issue_slab = [&]() {
      const unsigned sbase = (unsigned)(s_c * g.kd + s_z) * (unsigned)(NG * 4);
      const unsigned bufo = (unsigned)(s_buf * kSlab * 16);
      const int n0 = s_nb * NB;
#pragma unroll
      for (int j = 0; j < (kSlabI + kProd - 1) / kProd; ++j) {
        const int piece = pwv + j * kProd;                     // wave-uniform
        if (piece < kSlabI) {
          const int it = piece * 64 + lane;
          const int nn = it & (NB - 1), tp = it / NB;
          const unsigned off = n0 + nn < g.Npad ? ((sbase + (unsigned)tp) * (unsigned)g.Npad + (unsigned)(n0 + nn)) * 32u : 0x80000000u;
          const unsigned mh = __builtin_amdgcn_readfirstlane(lds_bhi + bufo + (unsigned)piece * 1024u);
          const unsigned ml = __builtin_amdgcn_readfirstlane(lds_blo + bufo + (unsigned)piece * 1024u);
          const unsigned off_lo = off + 16u;                   // (bit 31 survives: out-of-range lanes stay out of range)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
                       "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, 0 offen lds"
                       :: "s"(mh), "s"(ml), "v"(off), "v"(off_lo), "s"(wsrs) : "memory");
          issued += 2;
        }
      }
      ++s_n;
      if (++s_buf == 3) s_buf = 0;
      if (++s_z == KD) {
        s_z = 0;
        if (++s_c == nch) {
          s_c = 0;
          if (++s_k < nit) { int b_, d_, h_, w_; decode(it_beg + s_k, b_, d_, h_, w_, s_nb); }
        }
      }
    };
    set_item(0);
    if constexpr (kFlags) {
      // FLAG-SYNCHRONISED (no workgroup barrier in the main loop).  A producer wave runs as far ahead as the ring and the
      // three slab buffers allow, given the slowest consumer (t_min = min of the consumers' completed-step counts): slab s
      // may overwrite slab s - 3 once t_min >= s - 2, plane P may overwrite plane P - R once P - R is dead
      // (dead = PD * (t_min / KD) + t_min % KD: the wave on output plane 0 of the tile reads patch plane z at window plane z).
      // What it issued in EARLIER passes has landed once vmcnt <= what THIS pass issued; then the wave publishes those
      // counts (release) and the consumers that wait for them go on.  The consumers drift apart by up to a step: while one
      // wave of a SIMD waits for its first fragments of a step, the other is still multiplying the previous one -- with
      // the barrier both waited together at the start of every step (1.2 k of a step's 4.1 k cycles, round 4).
      int pub_p = 0, pub_s = 0, spins = 0;
      while (true) {
        int tmin = __hip_atomic_load(f_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 1; i < kCons; ++i) tmin = min(tmin, __hip_atomic_load(f_done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        tmin = __builtin_amdgcn_readfirstlane(tmin);
        const int iP0 = iP, sn0 = s_n;
        issued = 0;
        while (s_n < total_steps && s_n <= tmin + 2 && issued < 20 && g.dbg != 6) issue_slab();
        const int dead = PD * (tmin / KD) + tmin % KD;
        const int limit = min(dead + R, total_planes);
        while (iP < limit && issued < 24 && g.dbg != 6) { issue_plane(); issued += 2; }
        if (g.dbg == 6) { iP = total_planes; s_n = total_steps; }
        wait_vmcnt_le(issued);
        if (iP0 != pub_p || sn0 != pub_s) {
          pub_p = iP0; pub_s = sn0;
          if (lane == 0) {
            __hip_atomic_store(f_planes + pwv, pub_p, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(f_slabs + pwv, pub_s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (pub_p == total_planes && pub_s == total_steps) break;
        if (issued == 0 && iP == pub_p && s_n == pub_s) {          // nothing to do until a consumer moves on
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 22) || __hip_atomic_load(f_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            if (lane == 0) __hip_atomic_store(f_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
        } else {
          spins = 0;
        }
      }
      return;
    }
    int G = 0, z = 0;                                          // step t = (global chunk G, window plane z)
    const bool pst = g.stamps && tid == kCons * 64;
    long long p_issue = 0, p_wait = 0, p_bar = 0, pm0 = 0, pm1 = 0, pm2 = 0;
    for (int t = 0; t <= total_steps; ++t) {
      if (pst) pm0 = (long long)__builtin_amdgcn_s_memtime();
      issued = 0;
      // slabs up to step t + 1 (iteration 0: steps 0 and 1)
      while (s_n < total_steps && s_n <= t + 1 && g.dbg != 6) issue_slab();
      // planes that no step from t - 1 on reads (the consumers multiply step t - 1 during this iteration)
      int dead = 0;
      if (t >= 2) {
        const int G2 = z >= 2 ? G : G - 1, z2 = z >= 2 ? z - 2 : z - 2 + KD;
        dead = z2 < KD - 1 ? PD * G2 + z2 + 1 : PD * (G2 + 1);
      }
      const int limit = min(dead + R, total_planes);
      while (iP < limit && g.dbg != 6) { issue_plane(); issued += 2; }
      if (pst) pm1 = (long long)__builtin_amdgcn_s_memtime();
      // everything issued BEFORE this iteration has landed at the barrier that closes it (the slab and the planes of step
      // t were issued in iteration t - 1 or earlier: a plane is at most 8 behind the dead count, R >= 9); what this
      // iteration issued stays in flight.  Iteration 0 waits for all of it: step 0 multiplies next.
      wait_vmcnt_le(t == 0 ? 0 : issued);
      if (pst) pm2 = (long long)__builtin_amdgcn_s_memtime();
      __syncthreads();
      if (pst) { const long long e = (long long)__builtin_amdgcn_s_memtime(); p_issue += pm1 - pm0; p_wait += pm2 - pm1; p_bar += e - pm2; }
      if (++z == KD) { z = 0; ++G; }
    }
    if (pst) { if (blockIdx.x == 0) { g.stamps[8] = p_issue; g.stamps[9] = p_wait; g.stamps[10] = p_bar; } g.stamps[16 + 4 * blockIdx.x + 2] = p_wait; }
    return;
  }

  // -------------------------------------------------- consumers --------------------------------------------------
  int toff[NG];
  int pa[kMSUB];
  f32x4 acc[kMSUB][NSUB];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    const int t = gq * 4 + kk < g.KHW ? gq * 4 + kk : 0;       // slots past the window carry zero weights
    const int zh = mdiv(t, g.magic_kw), zw = t - zh * g.kw;
    toff[gq] = zh * kPW + zw;
  }
  const int sd_w = wave >> 1;                                  // the wave's output plane of the 4 x 8 x 16 tile
#pragma unroll
  for (int ms = 0; ms < kMSUB; ++ms) pa[ms] = ((wave & 1) * 4 + ms) * kPW + i16;
#pragma unroll
  for (int ms = 0; ms < kMSUB; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // step t - 1 (multiplied in iteration t): item k, chunk c, window plane z; pslot = ring slot of plane 0 of its chunk
  int k = 0, c = 0, z = 0, pslot = 0, sbuf = 0;                // (sbuf = slab buffer of the step = step % 3)
  int b, d0, h0, w0, nb;
  decode(it_beg, b, d0, h0, w0, nb);
  const bool cst = g.stamps && tid == 0;
  const bool wst = g.stamps && lane == 0 && blockIdx.x == 0;      // every consumer wave of workgroup 0: multiply / barrier cycles
  long long w_m = 0, w_b = 0, wm0 = 0, wm1 = 0;
  long long c_dma = 0, c_mfma = 0, c_wait = 0, c_bar = 0, cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0, c_start = 0, c_rt0 = 0;
  if (cst) { c_start = (long long)__builtin_amdgcn_s_memtime(); c_rt0 = (long long)__builtin_amdgcn_s_memrealtime(); }
  auto multiply_step = [&]() {
    {
      const int zr = __builtin_amdgcn_readfirstlane(zrtab[nb * nch + c]);
      if (z >= (zr & 255) && z < (zr >> 8) && g.dbg != 1 && !((g.dbg == 7 || g.dbg == 11) && wave >= 4) && !(g.dbg == 8 && wave < 4)) {    // (outside: only structural zeros)
        int slot = pslot + z + sd_w;
        if (slot >= R) slot -= R;
        if (slot >= R) slot -= R;
        const int abase = slot * kPL;
        const bf16x8* bh0 = Bhi + sbuf * kSlab;
        const bf16x8* bl0 = Blo + sbuf * kSlab;
        if constexpr (kSlide) {
          // ROW-SLIDING tap order, TAP-ROW-MAJOR.  The wave's four sub-tiles are four consecutive H rows of one plane, and
          // with the lane groups on the zw taps (kk = zw) the A fragment of (sub-tile m, tap row q) is patch row m + q of
          // the wave: the 4 x NGS (sub-tile, tap row) pairs read only 4 + NGS - 1 distinct rows.  Iteration q multiplies
          // rows q .. q + 3 with the B fragments of tap row q: 12 NSUB MFMAs on 4 NSUB independent accumulators (hi.hi of
          // all, then hi.lo of all, then lo.hi of all: a dependent MFMA is never closer than four behind its predecessor)
          // for ONE new row and ONE new B fragment pair, both requested kAhead iterations earlier -- every iteration has the
          // same shape, there is no ramp with one or two accumulators at the ends of a step (the row-major order of the
          // first version: rows 0 and NR - 1 meet a single tap row, three dependent MFMAs back to back; a lone wave issued
          // one MFMA per 31 cycles).  46 fragment reads per step for the 5 x 5 window planes instead of 70 (the zw = 4
          // column goes through two ordinary groups), 22 instead of 40 for the 4 x 4 ones.  The weights stay in the slab's
          // tap order; only the slot -> tap map of the fragment reads changes.  Per accumulator the products come in the
          // order tap row 0 .. NGS - 1, then the two zw = 4 groups.
          constexpr int NGS = NG == 7 ? 5 : 4, NR = kMSUB + NGS - 1;
          constexpr int kAhead = NSUB == 1 ? 2 : 1, NRB = kMSUB + kAhead, NBB = kAhead + 1;   // (an iteration is 12 NSUB MFMAs)
          constexpr int qA = NGS - 1;                          // iteration that requests the fragments of the first zw = 4 group
          constexpr int kw_ = NG == 7 ? 5 : 4;
          const unsigned brow = (unsigned)(kk * NB + i16);
          // (the row pitch is a compile-time constant: a row, a tap row, a sub-tile are IMMEDIATE offsets of the LDS reads --
          // with a run-time pitch the 46 reads of a step carried 57 address VALU instructions next to its 84 MFMAs, and a wave
          // alone issued one MFMA per 30 cycles instead of 17, round 4)
          const unsigned arow = (unsigned)(abase + ((wave & 1) * 4) * kPW + i16 + kk);     // row 0 of the wave, this lane's zw tap
          bf16x8 rh[NRB], rl[NRB], bsh[NBB][NSUB], bsl[NBB][NSUB];
          const bool rd = g.dbg != 9 && g.dbg != 11;             // (ablation: no fragment reads)
          auto load_row = [&](int r) { if (rd) { rh[r % NRB] = Ahi[arow + (unsigned)(r * kPW)]; rl[r % NRB] = Alo[arow + (unsigned)(r * kPW)]; } };
          auto load_b = [&](int q) {                           // slot (zh = q, zw = kk) = tap q * kw + kk of the slab
            if (rd)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
              bsh[q % NBB][ns] = bh0[brow + (unsigned)(q * kw_ * NB + ns * 16)];
              bsl[q % NBB][ns] = bl0[brow + (unsigned)(q * kw_ * NB + ns * 16)];
            }
          };
          // the zw = 4 column of the 5 x 5 window: group A = (zh = kk, zw = 4), group B = (zh = 4, zw = 4) + three zero slots
          bf16x8 th[kMSUB], tl[kMSUB], tbh[NSUB], tbl[NSUB], uh[kMSUB], ul[kMSUB], ubh[NSUB], ubl[NSUB];
          const unsigned tap5n = (unsigned)((NG == 7 ? (kk * 5 + 4) : 0) * NB + i16), tap6n = (unsigned)((NG == 7 ? 24 + kk : 0) * NB + i16);
          const unsigned rowA = arow + (unsigned)(kk * (kPW - 1) + 4), rowB = arow + (unsigned)(4 * kPW + 4) - (unsigned)kk;
          auto load_ga = [&]() {
            if (!rd) return;
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) { tbh[ns] = bh0[tap5n + (unsigned)(ns * 16)]; tbl[ns] = bl0[tap5n + (unsigned)(ns * 16)]; }
#pragma unroll
            for (int m = 0; m < kMSUB; ++m) { th[m] = Ahi[rowA + (unsigned)(m * kPW)]; tl[m] = Alo[rowA + (unsigned)(m * kPW)]; }
          };
          auto load_gb = [&]() {
            if (!rd) return;
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) { ubh[ns] = bh0[tap6n + (unsigned)(ns * 16)]; ubl[ns] = bl0[tap6n + (unsigned)(ns * 16)]; }
#pragma unroll
            for (int m = 0; m < kMSUB; ++m) { uh[m] = Ahi[rowB + (unsigned)(m * kPW)]; ul[m] = Alo[rowB + (unsigned)(m * kPW)]; }
          };
#pragma unroll
          for (int q = 0; q < kAhead; ++q) load_b(q);
#pragma unroll
          for (int r = 0; r < kMSUB - 1 + kAhead; ++r) load_row(r);
#pragma unroll
          for (int q = 0; q < NGS; ++q) {
            if (q + kAhead + kMSUB - 1 < NR) load_row(q + kAhead + kMSUB - 1);
            if (q + kAhead < NGS) load_b(q + kAhead);
            if constexpr (NG == 7) {
              if (NSUB == 1 && q == qA) load_ga();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
              for (int m = 0; m < kMSUB; ++m)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) {
                  f32x4& a = acc[m][ns];
                  const bf16x8& av = pass == 2 ? rl[(q + m) % NRB] : rh[(q + m) % NRB];
                  const bf16x8& bv = pass == 1 ? bsl[q % NBB][ns] : bsh[q % NBB][ns];
                  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (NG == 7) {
            if (NSUB != 1) { load_ga(); __builtin_amdgcn_sched_barrier(0); }
            if (NSUB == 1) load_gb();                          // (in flight under the MFMAs of the first group; with 32 columns
            __builtin_amdgcn_sched_barrier(0);                 // the registers of both groups at once do not fit 168)
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
              for (int m = 0; m < kMSUB; ++m)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) {
                  f32x4& a = acc[m][ns];
                  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 2 ? tl[m] : th[m], pass == 1 ? tbl[ns] : tbh[ns], a, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (NSUB != 1) { load_gb(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
#pragma unroll
              for (int m = 0; m < kMSUB; ++m)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) {
                  f32x4& a = acc[m][ns];
                  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 2 ? ul[m] : uh[m], pass == 1 ? ubl[ns] : ubh[ns], a, 0, 0, 0);
                }
          }
        } else {
        // software pipeline over "units" u = (tap group gq, half of the wave's sub-tiles), as in conv_bf3_ws_kernel
        constexpr int MH = NSUB == 1 ? kMSUB : kMSUB / 2, HPG = kMSUB / MH, NU = NG * HPG;
        bf16x8 bh[2][NSUB], bl[2][NSUB], ah[2][MH], al[2][MH];
        auto fragsB = [&](int gq, int qq) {
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) {
            bh[qq][ns] = bh0[(gq * 4 + kk) * NB + ns * 16 + i16];
            bl[qq][ns] = bl0[(gq * 4 + kk) * NB + ns * 16 + i16];
          }
        };
        auto fragsA = [&](int u, int qq) {
          const int gq = u / HPG, hf = u - gq * HPG;
          const int off = toff[gq] + abase;
#pragma unroll
          for (int m = 0; m < MH; ++m) { ah[qq][m] = Ahi[pa[hf * MH + m] + off]; al[qq][m] = Alo[pa[hf * MH + m] + off]; }
        };
        fragsB(0, 0);
        fragsA(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int gq = u / HPG, hf = u - gq * HPG, qq = u & 1, qb = gq & 1;
          if (u + 1 < NU) {
            if (hf == HPG - 1) fragsB(gq + 1, qb ^ 1);
            fragsA(u + 1, qq ^ 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
              f32x4& a = acc[hf * MH + m][ns];
              // the three products as one block of adjacent MFMAs (see mfma3 in conv_bf3.hip)
              asm("v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
                  : "+v"(a) : "v"(ah[qq][m]), "v"(al[qq][m]), "v"(bh[qb][ns]), "v"(bl[qb][ns]));
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
        }
      if (z == KD - 1 && c == nch - 1) {
        // the item is complete: D row = kk*4 + r = W position of the sub-tile, col = i16 = channel
        float* yb = g.y.base + (int64_t)b * g.y.sB;
        const int n0 = nb * NB;
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          const int n = n0 + ns * 16 + i16;
          if (n < g.y.C) {
            const int64_t co = view_chan(g.y, n);
            const float bsv = g.bias ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
#pragma unroll
            for (int ms = 0; ms < kMSUB; ++ms) {
              const int od = d0 + sd_w, oh = h0 + (wave & 1) * 4 + ms, ow = w0 + kk * 4;
              if (od < g.y.D && oh < g.y.H && ow < g.y.W) {
                float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
                if (g.vec_store) {
                  f32x4 v = acc[ms][ns] + bsv;
                  if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
                  *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    if (ow + r < g.y.W) {
                      float* dd = dst + (int64_t)r * g.y.sW;
                      const float v = acc[ms][ns][r] + bsv;
                      *dd = g.mode == 1 ? *dd + v : v;
                    }
                  }
                }
              }
            }
          }
        }
#pragma unroll
        for (int ms = 0; ms < kMSUB; ++ms)
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // advance to the next step
      if (++sbuf == 3) sbuf = 0;
      if (++z == KD) {
        z = 0;
        pslot += PD; if (pslot >= R) pslot -= R;
        if (++c == nch) {
          c = 0;
          if (++k < nit) decode(it_beg + k, b, d0, h0, w0, nb);
        }
      }
    }
  };
  if constexpr (kFlags) {
    // step t needs: slab t, and patch plane z + sd_w of its chunk = plane PD * (chunk count) + z + sd_w in issue order
    int need = sd_w, zc = 0;
    for (int t = 0; t < total_steps; ++t) {
      if (wst) wm0 = (long long)__builtin_amdgcn_s_memtime();
      int spins = 0;
      while (true) {
        int fp = __hip_atomic_load(f_planes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int fs = __hip_atomic_load(f_slabs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 1; i < kProd; ++i) {
          fp = min(fp, __hip_atomic_load(f_planes + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          fs = min(fs, __hip_atomic_load(f_slabs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
        if (__builtin_amdgcn_readfirstlane((fp > need && fs > t) ? 1 : 0)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22) || __hip_atomic_load(f_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          if (lane == 0) __hip_atomic_store(f_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          return;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (wst) wm1 = (long long)__builtin_amdgcn_s_memtime();
      multiply_step();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the step's LDS reads are complete: the MFMAs consumed them)
      if (lane == 0) __hip_atomic_store(f_done + wave, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wst) { const long long e = (long long)__builtin_amdgcn_s_memtime(); w_m += e - wm1; w_b += wm1 - wm0; }
      if (++zc == KD) { zc = 0; need += PD - (KD - 1); } else ++need;
    }
    if (wst) { g.stamps[16 + 4 * 256 + 2 * wave] = w_m; g.stamps[16 + 4 * 256 + 2 * wave + 1] = w_b; }
    if (cst) {
      g.stamps[16 + 4 * blockIdx.x + 0] = (long long)__builtin_amdgcn_s_memtime() - c_start;
      g.stamps[16 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memrealtime();
      if (blockIdx.x == 0) {
        g.stamps[0] = (long long)__builtin_amdgcn_s_memtime() - c_start;
        g.stamps[1] = (long long)__builtin_amdgcn_s_memrealtime() - c_rt0;
        g.stamps[6] = total_steps; g.stamps[7] = c_rt0;
      }
    }
    return;
  }
  for (int t = 0; t <= total_steps; ++t) {
    if (cst) cm0 = (long long)__builtin_amdgcn_s_memtime();
    if (cst) cm1 = (long long)__builtin_amdgcn_s_memtime();
    if (wst) wm0 = (long long)__builtin_amdgcn_s_memtime();
    if (t >= 1) multiply_step();
    if (cst) cm2 = (long long)__builtin_amdgcn_s_memtime();
    if (cst) cm3 = (long long)__builtin_amdgcn_s_memtime();
    if (wst) wm1 = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();
    if (wst) { const long long e = (long long)__builtin_amdgcn_s_memtime(); w_m += wm1 - wm0; w_b += e - wm1; }
    if (cst) { const long long e = (long long)__builtin_amdgcn_s_memtime(); c_dma += cm1 - cm0; c_mfma += cm2 - cm1; c_wait += cm3 - cm2; c_bar += e - cm3; }
  }
  if (wst) { g.stamps[16 + 4 * 256 + 2 * wave] = w_m; g.stamps[16 + 4 * 256 + 2 * wave + 1] = w_b; }
  if (cst) {
    g.stamps[16 + 4 * blockIdx.x + 0] = (long long)__builtin_amdgcn_s_memtime() - c_start;
    g.stamps[16 + 4 * blockIdx.x + 1] = c_bar;
    g.stamps[16 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (cst && blockIdx.x == 0) {
    g.stamps[0] = (long long)__builtin_amdgcn_s_memtime() - c_start;
    g.stamps[1] = (long long)__builtin_amdgcn_s_memrealtime() - c_rt0;
    g.stamps[2] = c_dma; g.stamps[3] = c_mfma; g.stamps[4] = c_wait; g.stamps[5] = c_bar; g.stamps[6] = total_steps; g.stamps[7] = c_rt0;
  }
}

template <int NSUB, int NG>
int launch_ring(const RingGeom& g, dim3 grid, size_t lds, hipStream_t st) {
  static const bool slide = getenv("CRN_RING_SLIDE") == nullptr || atoi(getenv("CRN_RING_SLIDE")) != 0;
  static const bool slide2 = getenv("CRN_RING_SLIDE2") == nullptr || atoi(getenv("CRN_RING_SLIDE2")) != 0;
  static const bool flags = getenv("CRN_RING_FLAGS") == nullptr || atoi(getenv("CRN_RING_FLAGS")) != 0;
  const bool sl = slide && (NSUB == 1 || slide2);
  auto k = flags ? (sl ? conv_bf3_ring_kernel<NSUB, NG, true, true> : conv_bf3_ring_kernel<NSUB, NG, false, true>)
                 : (sl ? conv_bf3_ring_kernel<NSUB, NG, true, false> : conv_bf3_ring_kernel<NSUB, NG, false, false>);
  CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(kRingThreads), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

unsigned magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }
long long* g_ring_stamps = nullptr;

}  // namespace

// tuning aid (CRN_RING_STAMPS=1): cycle sums of workgroup 0 of the last crn_conv_fwd_bf3_ring launch: consumer wave 0
// [0..6] = total shader cycles, total 100 MHz ticks, slab DMA issue, multiply (+ epilogue), slab wait, barrier, steps;
// producer wave 8 [8..10] = plane DMA issue, counted wait, barrier
extern "C" int crn_ring_debug_stamps(long long* out16) {   // (+ 4 per workgroup from [16] on: 16 + 4 * 256 in all)
  if (!g_ring_stamps) return CRN_EINVAL;
  CRN_HIP(hipDeviceSynchronize());
  CRN_HIP(hipMemcpy(out16, g_ring_stamps, (16 + 4 * 256 + 16) * sizeof(long long), hipMemcpyDeviceToHost));
  return CRN_OK;
}

extern "C" size_t crn_bf3_act_image_bytes(int B, int C, int D, int H, int W) {
  return (size_t)2 * B * ((C + kCK - 1) / kCK) * D * H * W * 16;
}

extern "C" int crn_bf3_act_image(const crnView* x, const crnInTransform* tr, void* image, crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !image || x->B <= 0 || x->C <= 0) return CRN_EINVAL;
  ImageGeom g{};
  g.x = *x;
  g.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
  g.NCH = (x->C + kCK - 1) / kCK;
  g.HW = x->H * x->W; g.S = x->D * g.HW;
  g.hi = reinterpret_cast<bf16x8*>(image);
  g.lo = g.hi + (size_t)x->B * g.NCH * g.S;
  if ((size_t)x->B * g.NCH * g.S * 16 >= ((size_t)1 << 31)) return CRN_EINVAL;     // (the conv kernel's 32-bit buffer offsets)
  dim3 grid((unsigned)crn_cdiv(g.S, 256), (unsigned)g.NCH, (unsigned)x->B);
  hipLaunchKernelGGL(bf3_act_image_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" size_t crn_bf3_ring_covers(int C, int Npad, int yD, int yH, int yW, int kd, int kh, int kw) {
  if (C <= 0 || Npad <= 0 || (Npad & 15)) return 0;
  if (!((kd == 5 && kh == 5 && kw == 5) || (kd == 4 && kh == 4 && kw == 4))) return 0;
  if (yW % 16 != 0 || yH < 8 || yD < 4) return 0;
  const int NB = Npad <= 16 ? 16 : 32;
  return ((Npad + NB - 1) / NB) * ((C + kCK - 1) / kCK) <= kHdr / 4 - 32 ? 1 : 0;     // (the last 32 words: the waves' sync flags)
}

// Returns CRN_EINVAL for shapes this kernel does not cover (the caller keeps crn_conv_fwd_bf3_slabs for those).
extern "C" int crn_conv_fwd_bf3_ring(const void* image, int B, int C, int D, int H, int W, const void* wslab, int Npad,
                                     const float* bias, int bias_sB, const crnView* y,
                                     int kd, int kh, int kw, int pd, int ph, int pw,
                                     int accumulate, const crnTapBoxes* boxes, crnStream stream) {
  CRN_ENTRY(stream);
  if (!image || !wslab || !y || Npad <= 0 || (Npad & 15) || B != y->B || C <= 0 || y->C > Npad) return CRN_EINVAL;
  if (boxes && (boxes->n_groups < 0 || boxes->n_groups > 8 || boxes->c_groups < 0 || boxes->c_groups > 8)) return CRN_EINVAL;
  if (!crn_bf3_ring_covers(C, Npad, y->D, y->H, y->W, kd, kh, kw)) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  RingGeom g{};
  g.B = B; g.C = C; g.NCH = (C + kCK - 1) / kCK; g.D = D; g.H = H; g.W = W;
  g.img_hi = image;
  g.img_lo = reinterpret_cast<const char*>(image) + (size_t)B * g.NCH * D * H * W * 16;
  if ((size_t)B * g.NCH * D * H * W * 16 >= ((size_t)1 << 31)) return CRN_EINVAL;
  g.y = *y; g.bias = bias; g.bias_sB = bias_sB; g.wslab = wslab; g.Npad = Npad;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.KHW = kh * kw;
  const int NG = (g.KHW + 3) / 4;
  g.PH = 8 + kh - 1; g.PW = 16 + kw - 1; g.PHW = g.PH * g.PW;       // (cubic windows: the kernels take PW = 16 + kd - 1 as a constant)
  if (g.PHW > kPL) return CRN_EINVAL;
  g.tilesD = crn_cdiv(y->D, 4); g.tilesH = crn_cdiv(y->H, 8); g.tilesW = y->W / 16;
  const int NSUB = Npad <= 16 ? 1 : 2, NB = NSUB * 16;
  g.NBK = crn_cdiv(Npad, NB);
  g.nch = g.NCH;
  if (g.NBK * g.nch > kHdr / 4 - 32) return CRN_EINVAL;
  const int64_t nitems = (int64_t)g.tilesD * g.tilesH * g.tilesW * B * g.NBK;
  if (nitems >= ((int64_t)1 << 24)) return CRN_EINVAL;
  g.nitems = (int)nitems;
  static const bool no_boxes = getenv("CRN_NO_BOXES") != nullptr;
  if (boxes && !no_boxes) {
    if (boxes->n_groups > 0 && y->C % boxes->n_groups == 0) { g.n_groups = boxes->n_groups; memcpy(g.n_box, boxes->n_box, sizeof(g.n_box)); }
    if (boxes->c_groups > 0 && C % boxes->c_groups == 0) { g.c_groups = boxes->c_groups; memcpy(g.c_box, boxes->c_box, sizeof(g.c_box)); }
  }
  const size_t slab_bytes = (size_t)6 * NG * 4 * NB * 16;        // 3 buffers x (hi + lo)
  int R = (int)((kLdsMax - kHdr - slab_bytes) / ((size_t)kPL * 32));
  static const int r_force = getenv("CRN_RING_PLANES") ? atoi(getenv("CRN_RING_PLANES")) : 0;
  R = std::min(R, r_force > 0 ? r_force : 12);
  if (R < 9) return CRN_EINVAL;
  g.R = R;
  const size_t lds = kHdr + (size_t)R * kPL * 32 + slab_bytes;
  g.mode = accumulate ? 1 : 0;
  const crnView& yo = g.y;
  g.vec_store = (yo.sW == 1 && (yo.W & 3) == 0 && (yo.sH & 3) == 0 && (yo.sD & 3) == 0 && (yo.sB & 3) == 0 &&
                 (yo.sC & 3) == 0 && (((uintptr_t)yo.base) & 15) == 0 && yo.chan_off == nullptr) ? 1 : 0;
  g.magic_kw = magic20(kw); g.magic_PW = magic20(g.PW);
  g.dbg = getenv("CRN_DBG_MODE") ? atoi(getenv("CRN_DBG_MODE")) : 0;
  static const bool want_stamps = getenv("CRN_RING_STAMPS") != nullptr;
  if (want_stamps) {
    if (!g_ring_stamps) CRN_HIP(hipMalloc(&g_ring_stamps, (16 + 4 * 256 + 16) * sizeof(long long)));
    CRN_HIP(hipMemsetAsync(g_ring_stamps, 0, (16 + 4 * 256 + 16) * sizeof(long long), st));
    g.stamps = g_ring_stamps;
  }
  static const int wgs = getenv("CRN_RING_WGS") ? atoi(getenv("CRN_RING_WGS")) : 256;     // one workgroup per CU
  g.items_per_wg = crn_cdiv(g.nitems, std::min(g.nitems, wgs));
  dim3 grid((unsigned)crn_cdiv(g.nitems, g.items_per_wg));
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_fwd_bf3_ring] x(C%d %dx%dx%d) y(C%d %dx%dx%d) k%d: NSUB %d NBK %d chunks %d items %d (%d per "
            "workgroup, %u workgroups) ring %d planes lds %zu\n", C, D, H, W, y->C, y->D, y->H, y->W, kd, NSUB, g.NBK, g.nch,
            g.nitems, g.items_per_wg, grid.x, R, lds);
  int rc = CRN_EINVAL;
#define CRN_RING_CASE(N, G) if (NSUB == N && NG == G) rc = launch_ring<N, G>(g, grid, lds, st);
  CRN_RING_CASE(1, 7) CRN_RING_CASE(2, 7) CRN_RING_CASE(1, 4) CRN_RING_CASE(2, 4)
#undef CRN_RING_CASE
  return rc;
}
