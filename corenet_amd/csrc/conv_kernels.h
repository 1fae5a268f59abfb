// Device code of the fp32 MFMA implicit-GEMM convolution engine (see conv_igemm.hip for the design).
// Kernel templates live here; conv_inst_*.hip instantiate them in parallel translation units
// (one hipcc process each) and export plain launcher functions used by conv_igemm.hip.
#pragma once
#include "crn_common.h"
#include <type_traits>

namespace crnk {

constexpr int PREG = 32;    // staged patch floats per thread   (CC*PS   <= 256*PREG)
constexpr int WREG = 8;     // staged weight float4 per thread  (CC*T*NB <= 256*WREG*4)
constexpr int DREG = 32;    // staged dy floats per thread      (NB*npos <= 256*DREG)
constexpr int kMaxStage = 256 * PREG;

struct ConvGeom {
  crnView x, y;
  crnInTransform tr;
  const float* w;
  const float* bias;
  int Npad, bias_sB;
  int kd, kh, kw, pd, ph, pw, T;
  int TD, TH, TW;          // tile (positions)
  int mw, mh;              // M-subtile shape, mw*mh == 16
  int nsh, nsw;            // sub-tiles per tile along H, W
  int PD, PH, PW, PS, PSP; // patch dims, size, padded channel stride
  int WSP;                 // LDS weight channel stride
  int CC;                  // channels per chunk (multiple of 4)
  int tilesD, tilesH, tilesW;
  int nchunks, chunks_per_split;
  int mode;                // 0 store, 1 accumulate (rmw), 2 atomic add, 3 split-K partials to scratch (+ reduce
                           // launch), 4 split-K partials to scratch, summed by the last workgroup of each tile
  crnView yreal;           // mode 4: the tensor the sums go to, accumulate_real: += instead of =
  int accumulate_real;
  int* counters;           // mode 4: one arrival counter per (tile, N block), zero between calls
  int lg2, npass;          // patch staging: plane padded to 2^lg2 slots, passes of 256 slots
  unsigned magic_PW, magic_PD, magic_T;
  int vec_store;           // epilogue may use 16-B stores (unit W stride, 4-aligned rows)
  int n_groups, c_groups;  // crnTapBoxes (0 = all taps)
  signed char n_box[8][6], c_box[8][6];
  int lead;                // extra patch columns on the left (16-byte staging), already included in pw/PW
  int nunits, plu, pw4;    // 16-byte staging geometry (PatchDesc)
  unsigned magic_PLU, magic_PW4;
  int dbg;                 // tuning aid (CRN_DBG_MODE): 1 = no MFMA loop, 2 = stage only the first chunk
};

__device__ __forceinline__ int64_t view_chan(const crnView& v, int c) {
  return v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
}

// ---- patch staging: global -> registers (issue) and registers -> LDS (commit) ----
// The patch of one chunk is CC*PD planes of PH*PW elements.  Staging slots are laid out as
// planes padded to PLP = 2^lg2 elements so that slot -> (plane q, in-plane r) is shift/mask:
//   lg2 >= 8: q is wave-uniform (scalar unit) and r takes PLP/256 per-thread values,
//   lg2 <  8: r is a per-thread constant.
// Everything that depends only on r (ph, pw, h/w bounds, h/w address part) is loop invariant
// and hoisted by the compiler; per slot ~6 VALU remain for issue and ~8 for commit.
struct PatchDesc {
  crnView x;
  crnInTransform tr;
  int pd, ph, pw, PD, PH, PW, plane, lg2, PSP;
  unsigned magic_PW, magic_PD;       // ceil(2^20 / d)
  // 16-byte staging (XV kernels): the patch is nunits float4 units, plu per plane, pw4 per row
  int nunits, plu, pw4;
  unsigned magic_PLU, magic_PW4;
};

// 16-byte staging.  Eligible views (unit W stride, everything 4-aligned) are staged as float4
// units: 4x fewer staging instructions, and on gfx950 every VALU/SALU instruction issued next to an
// MFMA stream costs MFMA issue time (tools/mfma_peak.hip).  The patch origin along W is moved
// left by `lead` (0..3) columns so that every row starts 16-byte aligned in global memory.
constexpr int NV = 8;       // float4 staging slots per thread (the same 32 registers as PREG)
constexpr int NVX = 9;      // ... for the x patch: 9 lets the 4x8x16-position tile of a k=5 conv (2304 units) fit

typedef int crn_rsrc __attribute__((ext_vector_type(4)));   // buffer resource (V#) in 4 SGPRs

__device__ __forceinline__ int mdiv(int x, unsigned magic) { return (int)(((unsigned)x * magic) >> 20); }

__device__ __forceinline__ crn_rsrc make_rsrc(const float* base) {   // raw buffer: stride 0, 2 GiB range
  const unsigned long long a = (unsigned long long)base;
  return (crn_rsrc){(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)0x7FFFFFFFu, 0x00020000};   // 2 GiB range: offsets with bit 31 set read 0
}

// Staging loads are inline asm on purpose: hipcc's waitcnt pass serialises compiler-visible loads
// whose destination VGPRs are re-used across the chunk loop (one s_waitcnt vmcnt(0) per load,
// measured 45k cycles per chunk).  The asm loads are invisible to that pass; the matching wait is
// crn_wait_loads() right before the registers are consumed (cdna guide 5.7).
// The destination is the staging register itself (no temporary): a compiler-inserted copy between
// the load and crn_wait_loads() would read the register before the data has landed.  For the same
// reason every staging slot is loaded unconditionally (inactive slots use the out-of-range offset).
// s_nop 4: the hazard recogniser does not look inside inline asm, and when the descriptor SGPRs were
// just rebuilt by VALU (v_readlane after an SGPR spill) a VMEM read needs 5 wait states (gfx9 ISA 4.5).
__device__ __forceinline__ void crn_bload(float& dst, const crn_rsrc& rs, unsigned byte_off) {
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rs));
}
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void crn_bload3(f32x3& dst, const crn_rsrc& rs, unsigned byte_off) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx3 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rs));
}
__device__ __forceinline__ void crn_bload4(f32x4& dst, const crn_rsrc& rs, unsigned byte_off) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rs));
}
template <int N>
__device__ __forceinline__ void crn_wait_loads(float (&v)[N]) {
  static_assert(N % 8 == 0, "");
#pragma unroll
  for (int i = 0; i < N; i += 8)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]), "+v"(v[i + 5]),
                   "+v"(v[i + 6]), "+v"(v[i + 7]));
}
template <int N>
__device__ __forceinline__ void crn_wait_loads4n(f32x3 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i]));
}
template <int N>
__device__ __forceinline__ void crn_wait_loads4n(f32x4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i]));
}
__device__ __forceinline__ void crn_wait_loads4(f32x4 (&v)[WREG]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// per-chunk channel tables in LDS: [0,64) element offsets, [64,128) BRN scale, [128,192) BRN shift
constexpr int kChTab = 192;
__device__ __forceinline__ void stage_choff(const crnView& v, const crnInTransform& tr, unsigned* choff, int c0,
                                            int nch) {
  const int t = threadIdx.x;
  if (t < nch) {
    const int c = min(c0 + t, v.C - 1);
    choff[t] = v.chan_off ? (unsigned)v.chan_off[c] : (unsigned)c * (unsigned)v.sC;
    if (tr.scale) {
      reinterpret_cast<float*>(choff)[64 + t] = tr.scale[c];
      reinterpret_cast<float*>(choff)[128 + t] = tr.shift[c];
    }
  }
}

// In-plane slot variant: everything that depends only on (thread, tile) and not on the plane.
// U = slots of 256 per padded plane (1, 2 or 4: compile time) -> U variants per thread; U == 0:
// planes smaller than 256 slots -> one variant, the plane index is per thread.
struct HWVar { int r; unsigned off; int in; };   // r = in-plane index (or -1), off = gh*sH + gw*sW

template <int U>
__device__ __forceinline__ void patch_hw(const PatchDesc& g, int h0, int w0, HWVar (&hw)[U ? U : 1]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < (U ? U : 1); ++u) {
    const int r = U ? tid + u * 256 : (tid & ((1 << g.lg2) - 1));
    const bool valid = r < g.plane;
    const int ph = mdiv(r, g.magic_PW), pw = r - ph * g.PW;
    const int gh = h0 + ph - g.ph, gw = w0 + pw - g.pw;
    hw[u].r = valid ? r : -1;
    hw[u].in = valid && (unsigned)gh < (unsigned)g.x.H && (unsigned)gw < (unsigned)g.x.W;
    hw[u].off = (unsigned)gh * (unsigned)g.x.sH + (unsigned)gw * (unsigned)g.x.sW;
  }
}

__device__ __forceinline__ unsigned chan_off_s(const crnView& v, int c) {   // c wave-uniform -> scalar load
  return v.chan_off ? (unsigned)v.chan_off[c] : (unsigned)c * (unsigned)v.sC;
}

template <int J, int U>
__device__ __forceinline__ void patch_issue_one(const PatchDesc& g, const unsigned* choff,
                                                const HWVar (&hw)[U ? U : 1], const crn_rsrc& rs, int nplanes,
                                                int c0, int d0, float& v) {
  unsigned goff = 0x80000000u;               // outside the descriptor range -> the load returns 0
  if constexpr (U >= 1) {
    int q = J / U;
    asm volatile("" : "+s"(q));              // recompute per chunk on the scalar unit
    const HWVar& h = hw[J % U];
    const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
    const int c = c0 + cl, gd = d0 + pd - g.pd;
    if (q < nplanes && c < g.x.C && (unsigned)gd < (unsigned)g.x.D) {       // wave-uniform
      // channel offset from the LDS table, NOT from global memory: a compiler-visible global load
      // here makes hipcc emit s_waitcnt vmcnt(0), which also drains every asm staging load issued so
      // far (measured: one full memory latency per slot, 27 us per wgrad tile)
      const unsigned sbase = choff[cl] + (unsigned)gd * (unsigned)g.x.sD;
      goff = h.in ? (sbase + h.off) * 4u : 0x80000000u;
    }
  } else {
    int jq = J * (256 >> g.lg2);
    asm volatile("" : "+s"(jq));
    const HWVar& h = hw[0];
    const int q = jq + ((int)threadIdx.x >> g.lg2);
    const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
    const int gd = d0 + pd - g.pd;
    const bool in = h.in && q < nplanes && (c0 + cl < g.x.C) && (unsigned)gd < (unsigned)g.x.D;
    if (in) goff = (choff[cl] + (unsigned)gd * (unsigned)g.x.sD + h.off) * 4u;
  }
  crn_bload(v, rs, goff);
}

template <int J, int U>
__device__ __forceinline__ void patch_commit_one(const PatchDesc& g, const unsigned* choff,
                                                 const HWVar (&hw)[U ? U : 1], float* ldsA, int nplanes, int c0,
                                                 int d0, float v) {
  if constexpr (U >= 1) {
    int q = J / U;
    asm volatile("" : "+s"(q));
    const HWVar& h = hw[J % U];
    if (q < nplanes) {                                   // wave-uniform
      const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
      const int c = c0 + cl, gd = d0 + pd - g.pd;
      if (g.tr.scale && c < g.x.C && (unsigned)gd < (unsigned)g.x.D) {
        const float* tab = reinterpret_cast<const float*>(choff);
        const float sc = tab[64 + cl], sh = tab[128 + cl];            // LDS broadcast reads
        float t = v;
        if (g.tr.pre_relu) t = fmaxf(t, 0.f);
        t = t * sc + sh;
        if (g.tr.post_relu) t = fmaxf(t, 0.f);
        v = h.in ? t : v;                                // zero padding stays zero
      }
      if (h.r >= 0) ldsA[cl * g.PSP + pd * g.plane + h.r] = v;
    }
  } else {
    int jq = J * (256 >> g.lg2);
    asm volatile("" : "+s"(jq));
    const HWVar& h = hw[0];
    const int q = jq + ((int)threadIdx.x >> g.lg2);
    if (q < nplanes && h.r >= 0) {
      const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
      const int gd = d0 + pd - g.pd;
      const bool in = h.in && (c0 + cl < g.x.C) && (unsigned)gd < (unsigned)g.x.D;
      if (in && g.tr.scale) {
        const float* tab = reinterpret_cast<const float*>(choff);
        if (g.tr.pre_relu) v = fmaxf(v, 0.f);
        v = v * tab[64 + cl] + tab[128 + cl];
        if (g.tr.post_relu) v = fmaxf(v, 0.f);
      }
      ldsA[cl * g.PSP + pd * g.plane + h.r] = v;
    }
  }
}

template <int U, int J = 0>
__device__ __forceinline__ void patch_issue_u(const PatchDesc& g, const unsigned* choff,
                                              const HWVar (&hw)[U ? U : 1], const crn_rsrc& rs, int nplanes,
                                              int npass, int c0, int d0, float (&val)[PREG]) {
  if constexpr (J < PREG) {
    patch_issue_one<J, U>(g, choff, hw, rs, J < npass ? nplanes : 0, c0, d0, val[J]);
    patch_issue_u<U, J + 1>(g, choff, hw, rs, nplanes, npass, c0, d0, val);
  }
}
template <int U, int J = 0>
__device__ __forceinline__ void patch_commit_u(const PatchDesc& g, const unsigned* choff,
                                               const HWVar (&hw)[U ? U : 1], float* ldsA, int nplanes, int npass,
                                               int c0, int d0, const float (&val)[PREG]) {
  if constexpr (J < PREG) {
    if (J < npass) patch_commit_one<J, U>(g, choff, hw, ldsA, nplanes, c0, d0, val[J]);
    patch_commit_u<U, J + 1>(g, choff, hw, ldsA, nplanes, npass, c0, d0, val);
  }
}

// Runtime (wave-uniform) dispatch on the plane padding; HWVar storage is sized for the largest U.
struct PatchHW { HWVar v[2]; };
__device__ __forceinline__ void patch_prepare(const PatchDesc& g, int h0, int w0, PatchHW& p) {
  switch (g.lg2) {
    case 8: patch_hw<1>(g, h0, w0, reinterpret_cast<HWVar(&)[1]>(p.v)); break;
    case 9: patch_hw<2>(g, h0, w0, reinterpret_cast<HWVar(&)[2]>(p.v)); break;
    default: patch_hw<0>(g, h0, w0, reinterpret_cast<HWVar(&)[1]>(p.v)); break;
  }
}
__device__ __forceinline__ void patch_issue(const PatchDesc& g, const unsigned* choff, const PatchHW& p,
                                            const crn_rsrc& rs, int nplanes, int npass, int c0, int d0,
                                            float (&val)[PREG]) {
  switch (g.lg2) {   // wave-uniform
    case 8: patch_issue_u<1>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), rs, nplanes, npass, c0, d0, val); break;
    case 9: patch_issue_u<2>(g, choff, reinterpret_cast<const HWVar(&)[2]>(p.v), rs, nplanes, npass, c0, d0, val); break;
    default: patch_issue_u<0>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), rs, nplanes, npass, c0, d0, val); break;
  }
}
__device__ __forceinline__ void patch_commit(const PatchDesc& g, const unsigned* choff, const PatchHW& p,
                                             float* ldsA, int nplanes, int npass, int c0, int d0,
                                             const float (&val)[PREG]) {
  switch (g.lg2) {
    case 8: patch_commit_u<1>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
    case 9: patch_commit_u<2>(g, choff, reinterpret_cast<const HWVar(&)[2]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
    default: patch_commit_u<0>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
  }
}

// unit u -> (channel cl, plane row pd, in-plane unit r); tile invariant, recomputed per use so that
// it does not sit in registers across the MFMA loop
struct VUnit { int cl, pd, r, ph, p4; bool valid; };
template <int J>
__device__ __forceinline__ VUnit vec_unit(const PatchDesc& g) {
  int u = (int)threadIdx.x + J * 256;
  asm volatile("" : "+v"(u));
  VUnit o;
  o.valid = u < g.nunits;
  const int q = mdiv(u, g.magic_PLU);
  o.r = u - q * g.plu;
  o.ph = mdiv(o.r, g.magic_PW4);
  o.p4 = o.r - o.ph * g.pw4;
  o.cl = mdiv(q, g.magic_PD);
  o.pd = q - o.cl * g.PD;
  return o;
}
// PU = positions per unit: 4 (unit-stride views, float4) or 2 (stride-2 space-to-depth views: dwordx3 load,
// elements 0 and 2 are two consecutive positions of this channel's parity; the patch origin is even, so both
// are inside or both outside, and the load ends exactly at the second one)
template <int PU, int J = 0, typename VT>
__device__ __forceinline__ void patch_issue_v(const PatchDesc& g, const unsigned* choff, const crn_rsrc& rs, int c0,
                                              int d0, int h0, int w0, VT (&val)[NVX], unsigned& inmask) {
  if constexpr (J < NVX) {
    unsigned goff = 0x80000000u;
    const VUnit t = vec_unit<J>(g);
    if (t.valid) {
      const int gd = d0 + t.pd - g.pd, gh = h0 + t.ph - g.ph, gw = w0 + PU * t.p4 - g.pw;
      const bool in = (c0 + t.cl < g.x.C) && (unsigned)gd < (unsigned)g.x.D && (unsigned)gh < (unsigned)g.x.H &&
                      (unsigned)gw < (unsigned)g.x.W;
      if (in) {
        goff = (choff[t.cl] + (unsigned)gd * (unsigned)g.x.sD + (unsigned)gh * (unsigned)g.x.sH +
                (unsigned)gw * (unsigned)(PU == 2 ? 2 : 1)) * 4u;
        inmask |= 1u << J;
      }
    }
    if constexpr (PU == 4) crn_bload4(val[J], rs, goff); else crn_bload3(val[J], rs, goff);
    patch_issue_v<PU, J + 1>(g, choff, rs, c0, d0, h0, w0, val, inmask);
  }
}
template <int PU, int J = 0, typename VT>
__device__ __forceinline__ void patch_commit_v(const PatchDesc& g, const unsigned* choff, float* ldsA,
                                               const VT (&val)[NVX], unsigned inmask) {
  if constexpr (J < NVX) {
    if (J * 256 < g.nunits) {                            // wave-uniform
      const VUnit t = vec_unit<J>(g);
      if (t.valid) {
        float v[PU];
#pragma unroll
        for (int e = 0; e < PU; ++e) v[e] = val[J][PU == 4 ? e : 2 * e];
        if (g.tr.scale && ((inmask >> J) & 1u)) {        // zero padding stays zero
          const float* tab = reinterpret_cast<const float*>(choff);
          const float sc = tab[64 + t.cl], sh = tab[128 + t.cl];
#pragma unroll
          for (int e = 0; e < PU; ++e) {
            float x = v[e];
            if (g.tr.pre_relu) x = fmaxf(x, 0.f);
            x = x * sc + sh;
            if (g.tr.post_relu) x = fmaxf(x, 0.f);
            v[e] = x;
          }
        }
        float* dst = ldsA + t.cl * g.PSP + t.pd * g.plane + PU * t.r;
        if constexpr (PU == 4) {
          *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<f32x2*>(dst) = (f32x2){v[0], v[1]};
        }
      }
    }
    patch_commit_v<PU, J + 1>(g, choff, ldsA, val, inmask);
  }
}

template <int V>
struct IntC { static constexpr int value = V; };

// Two workgroups share a CU and run the same [stage | MFMA] cycle; started together they stay in
// phase and fight for the same unit in every phase.  Delaying the workgroup in the odd CU slot by
// about half a cycle puts them in anti-phase: one stages while the other owns the MFMA pipe.
__device__ __forceinline__ void phase_skew(int sleeps) {
  const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);   // HW_ID.TG_ID
  if (tg & 1)
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
}

// XCD-aware work-item order (cdna guide T1): workgroup b is dispatched to XCD b % 8, each XCD has
// its own L2; remap so that each XCD walks a CONTIGUOUS range of tiles and the halos shared by
// neighbouring tiles hit in that XCD's L2 instead of being re-fetched from HBM by another one.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One chunk's MFMAs.  A (k-step, zd, zh) row of KW taps is straight-line code: the tap offsets
// along W are ds_read immediates, so a row costs MSUB+1 address adds for KW*MSUB*NSUB MFMAs.
// nzw > 1 (with KW == 1): window widths without an unrolled variant walk the taps one by one.
// tap box (half open) of the window that can hold non-zero weights
struct TapBox { int d0, d1, h0, h1, w0, w1; };
__host__ __device__ __forceinline__ TapBox box_union(const signed char (*box)[6], int groups, int nchan, int lo, int hi,
                                            int kd, int kh, int kw) {
  TapBox b{0, kd, 0, kh, 0, kw};
  if (groups <= 0) return b;
  const int per = max(1, nchan / groups);
  const int g0 = min(lo / per, groups - 1), g1 = min(hi / per, groups - 1);
  b = TapBox{kd, 0, kh, 0, kw, 0};
  for (int g = g0; g <= g1; ++g) {
    b.d0 = min(b.d0, (int)box[g][0]); b.d1 = max(b.d1, (int)box[g][1]);
    b.h0 = min(b.h0, (int)box[g][2]); b.h1 = max(b.h1, (int)box[g][3]);
    b.w0 = min(b.w0, (int)box[g][4]); b.w1 = max(b.w1, (int)box[g][5]);
  }
  return b;
}
__host__ __device__ __forceinline__ TapBox box_intersect(const TapBox& a, const TapBox& b) {
  return TapBox{max(a.d0, b.d0), min(a.d1, b.d1), max(a.h0, b.h0), min(a.h1, b.h1), max(a.w0, b.w0), min(a.w1, b.w1)};
}

template <int KW, int MSUB, int NSUB>
__device__ __forceinline__ void mfma_rows(f32x4 (&acc)[MSUB][NSUB], const float* ldsA, const float* ldsB,
                                          const int (&posbase)[MSUB], int bbase, int ksteps, const ConvGeom& g,
                                          int nzw, const TapBox& tb) {
  constexpr int NB = NSUB * 16;
  const float* pa0[MSUB];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) pa0[ms] = ldsA + posbase[ms];
  const float* pb0 = ldsB + bbase;
  for (int ks = 0; ks < ksteps; ++ks)
    for (int zd = tb.d0; zd < tb.d1; ++zd)
      for (int zh = tb.h0; zh < tb.h1; ++zh)
        for (int z0 = 0; z0 < nzw; ++z0) {
          const int aoff = ks * 4 * g.PSP + (zd * g.PH + zh) * g.PW + tb.w0 + z0;
          const int boff = ks * 4 * g.WSP + ((zd * g.kh + zh) * g.kw + tb.w0 + z0) * NB;
          const float* pb = pb0 + boff;
          float a[KW][MSUB], bv[KW][NSUB];
#pragma unroll
          for (int zw = 0; zw < KW; ++zw) {
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms) a[zw][ms] = pa0[ms][aoff + zw];
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) bv[zw][ns] = pb[zw * NB + ns * 16];
          }
#pragma unroll
          for (int zw = 0; zw < KW; ++zw)
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
              for (int ns = 0; ns < NSUB; ++ns)
                acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[zw][ms], bv[zw][ns], acc[ms][ns], 0, 0, 0);
          // issue order: reads run one tap ahead of the MFMAs that consume them
          __builtin_amdgcn_sched_group_barrier(0x100, MSUB + NSUB, 0);
#pragma unroll
          for (int zw = 0; zw < KW; ++zw) {
            __builtin_amdgcn_sched_group_barrier(0x100, MSUB + NSUB, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MSUB * NSUB, 0);
          }
        }
}

// ------------------------------- forward -----------------------------------
template <int MSUB, int NSUB, int XV>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvGeom g) {
  crn_kernarg_touch(g);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);      // 2 x kChTab per-chunk channel tables
  float* ldsA = lds + 2 * kChTab;
  float* ldsB = ldsA + g.CC * g.PSP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * g.TD, h0 = thi * g.TH, w0 = twi * g.TW;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int cbeg = split * g.chunks_per_split;
  const int cend = min(cbeg + g.chunks_per_split, g.nchunks);

  PatchDesc pdsc;
  pdsc.x = g.x; pdsc.tr = g.tr; pdsc.pd = g.pd; pdsc.ph = g.ph; pdsc.pw = g.pw;
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.plane = g.PH * g.PW; pdsc.lg2 = g.lg2;
  pdsc.PSP = g.PSP; pdsc.magic_PW = g.magic_PW; pdsc.magic_PD = g.magic_PD;
  pdsc.nunits = g.nunits; pdsc.plu = g.plu; pdsc.pw4 = g.pw4; pdsc.magic_PLU = g.magic_PLU; pdsc.magic_PW4 = g.magic_PW4;
  const int nplanes = g.CC * g.PD, npass = g.npass;
  const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
  PatchHW phw;
  if constexpr (XV == 0) patch_prepare(pdsc, h0, w0, phw);

  // lane's LDS offset of output position (sub-tile s, row i16) at tap (0,0,0)
  int posbase[MSUB];
  const int ri = i16 / g.mw, rj = i16 - ri * g.mw;
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) {
    int s = wave * MSUB + ms;
    const int sw = s % g.nsw; s /= g.nsw;
    const int sh = s % g.nsh; s /= g.nsh;
    const int sd = s;
    posbase[ms] = (sd * g.PH + sh * g.mh + ri) * g.PW + sw * g.mw + rj + kk * g.PSP + g.lead;
  }
  const int bbase = kk * g.WSP + i16;

  f32x4 acc[MSUB][NSUB];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const TapBox nbox = box_union(g.n_box, g.n_groups, g.y.C, n0, min(n0 + NB, g.y.C) - 1, g.kd, g.kh, g.kw);
  constexpr int XPU = XV == 2 ? 2 : 4;                  // positions per staged unit
  using XVT = typename std::conditional<XV == 2, f32x3, f32x4>::type;
  float pval[XV ? 1 : PREG];
  XVT pv4[XV ? NVX : 1];
  unsigned inmask = 0;
  f32x4 wval[WREG];
  const int nf4 = g.CC * g.T * (NB / 4);

  const crn_rsrc wrs = make_rsrc(g.w);
  // weight element f (float4) -> (LDS offset, global float offset or -1)
  auto weight_elem = [&](int f, int c0, int& ldso, int64_t& go) {
    const int j4 = f % (NB / 4);
    const int ct = f / (NB / 4);
    const int cl = mdiv(ct, g.magic_T);
    const int t = ct - cl * g.T;
    const int c = c0 + cl;
    const int n = n0 + j4 * 4;
    ldso = cl * g.WSP + t * NB + j4 * 4;
    go = (c < g.x.C && n < g.Npad) ? ((int64_t)c * g.T + t) * g.Npad + n : -1;
  };
  auto weights_issue = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WREG; ++j) {
      int f = tid + j * 256;
      asm volatile("" : "+v"(f));
      unsigned boff = 0x80000000u;             // out of range -> zeros
      if (f < nf4) {
        int ldso; int64_t go;
        weight_elem(f, c0, ldso, go);
        if (go >= 0) boff = (unsigned)go * 4u;
      }
      crn_bload4(wval[j], wrs, boff);
    }
  };
  auto weights_commit = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WREG; ++j) {
      int f = tid + j * 256;
      asm volatile("" : "+v"(f));
      if (f < nf4) {
        int ldso; int64_t go;
        weight_elem(f, c0, ldso, go);
        *reinterpret_cast<f32x4*>(ldsB + ldso) = wval[j];
      }
    }
  };

  if (cbeg < cend) {
    stage_choff(g.x, g.tr, choff + (cbeg & 1) * kChTab, cbeg * g.CC, g.CC);
    __syncthreads();
    if constexpr (XV != 0) patch_issue_v<XPU>(pdsc, choff + (cbeg & 1) * kChTab, xrs, cbeg * g.CC, d0, h0, w0, pv4, inmask);
    else patch_issue(pdsc, choff + (cbeg & 1) * kChTab, phw, xrs, nplanes, npass, cbeg * g.CC, d0, pval);
    weights_issue(cbeg * g.CC);
  }
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int c0 = chunk * g.CC;
    const bool stage = !(g.dbg >= 2 && chunk > cbeg);
    if (stage) {
    if constexpr (XV != 0) crn_wait_loads4n(pv4); else crn_wait_loads(pval);
    crn_wait_loads4(wval);
    __syncthreads();                       // previous chunk's MFMA reads are done
    if constexpr (XV != 0) { patch_commit_v<XPU>(pdsc, choff + (chunk & 1) * kChTab, ldsA, pv4, inmask); inmask = 0; }
    else patch_commit(pdsc, choff + (chunk & 1) * kChTab, phw, ldsA, nplanes, npass, c0, d0, pval);
    weights_commit(c0);
    if (chunk + 1 < cend) stage_choff(g.x, g.tr, choff + ((chunk + 1) & 1) * kChTab, c0 + g.CC, g.CC);
    __syncthreads();
    }
    if (stage && g.dbg < 2 && chunk + 1 < cend) {                // next chunk's loads fly under this chunk's MFMAs
      if constexpr (XV != 0) patch_issue_v<XPU>(pdsc, choff + ((chunk + 1) & 1) * kChTab, xrs, c0 + g.CC, d0, h0, w0, pv4, inmask);
      else patch_issue(pdsc, choff + ((chunk + 1) & 1) * kChTab, phw, xrs, nplanes, npass, c0 + g.CC, d0, pval);
      weights_issue(c0 + g.CC);
    }

    // MFMA loop (mfma_rows): the switch on the tap-row width sits OUTSIDE the loops, so the
    // accumulators never cross a control-flow merge inside them (a merge costs 32 v_mov per row,
    // and on gfx950 every VALU instruction next to an MFMA stream costs 4-8 MFMA cycles:
    // tools/mfma_peak.hip).
    if (g.dbg != 1) {
      const int ksteps = g.CC >> 2;
      // taps that can hold non-zero weights for this block's output columns and this chunk's channels
      const TapBox tb = box_intersect(nbox, box_union(g.c_box, g.c_groups, g.x.C, c0, min(c0 + g.CC, g.x.C) - 1,
                                                      g.kd, g.kh, g.kw));
      if (tb.d1 > tb.d0 && tb.h1 > tb.h0 && tb.w1 > tb.w0)
      switch (tb.w1 - tb.w0) {
        case 1: mfma_rows<1, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        case 2: mfma_rows<2, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        case 3: mfma_rows<3, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        case 4: mfma_rows<4, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        case 5: mfma_rows<5, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        case 7: mfma_rows<7, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, 1, tb); break;
        default: mfma_rows<1, MSUB, NSUB>(acc, ldsA, ldsB, posbase, bbase, ksteps, g, tb.w1 - tb.w0, tb); break;
      }
    }
  }

  // epilogue: D row = kk*4 + r (position), col = i16 (channel)
  // mode 3: split-K partial sums go to a dense scratch tensor [split][b][n][pos] (plain stores); a second
  // launch adds them up.  Device-scope float atomics (mode 2) leave the XCD's L2 and cost ~30 us per launch.
  float* yb = g.y.base + (int64_t)(g.mode >= 3 ? split * g.x.B + b : b) * g.y.sB;
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    const int n = n0 + ns * 16 + i16;
    if (n >= g.y.C) continue;
    const int64_t co = view_chan(g.y, n);
    const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      int s = wave * MSUB + ms;
      const int sw = s % g.nsw; s /= g.nsw;
      const int sh = s % g.nsh; s /= g.nsh;
      const int sd = s;
      if (g.vec_store) {
        // the 4 accumulator rows of this lane are 4 consecutive W positions: one 16-B store
        const int row0 = kk * 4;
        const int rr = row0 / g.mw, rc = row0 - rr * g.mw;
        const int od = d0 + sd, oh = h0 + sh * g.mh + rr, ow = w0 + sw * g.mw + rc;
        if (od < g.y.D && oh < g.y.H && ow < g.y.W) {
          float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + ow;
          f32x4 v = acc[ms][ns] + bsv;
          if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row_ = kk * 4 + r;
        const int rr = row_ / g.mw, rc = row_ - rr * g.mw;
        const int od = d0 + sd, oh = h0 + sh * g.mh + rr, ow = w0 + sw * g.mw + rc;
        if (od < g.y.D && oh < g.y.H && ow < g.y.W) {
          float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
          const float v = acc[ms][ns][r] + bsv;
          if (g.mode == 0 || g.mode >= 3) *dst = v;
          else if (g.mode == 1) *dst += v;
          else atomicAdd(dst, v);
        }
      }
    }
  }
  // mode 4: the split-K reduction without a second launch.  Every workgroup of a tile publishes its partial sums
  // (release fence), takes a ticket, and the one that arrives last adds up all splits in a fixed order (split 0
  // first: the result does not depend on who arrives last) and writes the real output.
  // MEASURED AND NOT USED BY DEFAULT (env CRN_SPLITK_FUSED=1 turns it on): the agent-scope release / acquire fences
  // every workgroup needs write back / invalidate the XCD's L2 (MI355X: 8 XCDs, L2 not coherent across them), which
  // cost far more than the 66 reduction launches (0.48 ms) they save: 18.7 ms per training step instead of 10.5.
  if (g.mode == 4) {
    __threadfence();
    int* s_last = reinterpret_cast<int*>(lds);            // the channel tables are dead by now
    __syncthreads();
    if (tid == 0) {
      const int id = blockIdx.x + gridDim.x * blockIdx.y;
      const int ticket = atomicAdd(g.counters + id, 1);
      const int last = ticket == (int)gridDim.z - 1;
      if (last) g.counters[id] = 0;                       // ready for the next call on this stream
      *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    __threadfence();
    const int64_t slab = (int64_t)g.x.B * g.y.sB;          // one split of the scratch tensor
    const int nsplit = gridDim.z;
    const float* sb = g.y.base + (int64_t)b * g.y.sB;
    float* yr = g.yreal.base + (int64_t)b * g.yreal.sB;
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int n = n0 + ns * 16 + i16;
      if (n >= g.y.C) continue;
      const int64_t co = view_chan(g.y, n), cor = view_chan(g.yreal, n);
#pragma unroll
      for (int ms = 0; ms < MSUB; ++ms) {
        int s = wave * MSUB + ms;
        const int sw = s % g.nsw; s /= g.nsw;
        const int sh = s % g.nsh; s /= g.nsh;
        const int sd = s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row_ = kk * 4 + r;
          const int rr = row_ / g.mw, rc = row_ - rr * g.mw;
          const int od = d0 + sd, oh = h0 + sh * g.mh + rr, ow = w0 + sw * g.mw + rc;
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

// ------------------------------ weight grad ---------------------------------
struct WgradGeom {
  crnView x, dy;
  crnInTransform tr;
  float* dw;
  int Npad;
  int kd, kh, kw, pd, ph, pw, T;
  int TD, TH, TW;
  int PD, PH, PW, PSP;
  int POSP;                // LDS stride of one dy channel ([n][pos] layout), npos + 4
  int lg2, npass;          // patch staging slots
  int dlg2, dnpass;        // dy staging slots (plane = TD*TH*TW positions of one channel)
  int CC;                  // channels per block (rows = CC*T <= 64*RSUB)
  int tilesD, tilesH, tilesW, ntiles;   // ntiles includes batch
  int tiles_per_split;
  unsigned magic_PW, magic_PD, magic_T, magic_TW, magic_TH;
  int lead;                // 16-byte staging of x (see ConvGeom)
  int nunits, plu, pw4;
  unsigned magic_PLU, magic_PW4;
  int xcd;                 // XCD-aware block order
  // tap sub-box of a larger packed window (crnTapBoxes): local tap (zd,zh,zw) is packed tap
  // ((zd+bd0)*khf + zh+bh0)*kwf + zw+bw0 of Tfull; ncols = valid columns of this launch
  int Tfull, khf, kwf, bd0, bh0, bw0, ncols;
  int n_groups;            // crnTapBoxes of the output columns (0 = all taps)
  signed char n_box[8][6];
  // balanced split counts: output group g (fewer real taps -> cheaper tiles) gets sg[g] splits of tps[g]
  // tiles; grid.z enumerates (group, n-block in group, split) through the prefix sums zoff[], grid.y = 1, so
  // that every launched block has work and all of them fit in the single resident round
  int balanced, nbpg;      // nbpg = n-blocks per group
  int tps[8], sg[8], zoff[9];
  int dnunits, np4;        // 16-byte staging of dy: NB * npos/4 units, np4 = npos/4 per channel
  unsigned magic_NP4;
  int dbg;                 // tuning aid (CRN_DBG_MODE): 1 = no MFMA loop, 2 = stage only the first tile
};

// One tile's MFMAs: a (td,th) row of WS = TW/4 k-steps is straight-line code.
// NACT <= RSUB: row sub-tiles of this wave that hold real (channel, tap) rows; the others are skipped
// (tap boxes of transposed convolutions leave up to 58 % of the 64-tap rows structurally zero).
template <int WS, int RSUB, int NSUB, int NACT>
__device__ __forceinline__ void wgrad_rows(f32x4 (&acc)[RSUB][NSUB], const float* ldsA, const float* ldsB,
                                           const int (&rowbase)[RSUB], int bbase, const WgradGeom& g) {
  const float* pa0[NACT];
#pragma unroll
  for (int rs = 0; rs < NACT; ++rs) pa0[rs] = ldsA + rowbase[rs];
  const float* pb0 = ldsB + bbase;
  for (int td = 0; td < g.TD; ++td)
    for (int th = 0; th < g.TH; ++th) {
      const int aoff = (td * g.PH + th) * g.PW;
      const float* pb = pb0 + (td * g.TH + th) * g.TW;
      float a[WS][NACT], bv[WS][NSUB];
#pragma unroll
      for (int ws = 0; ws < WS; ++ws) {
#pragma unroll
        for (int rs = 0; rs < NACT; ++rs) a[ws][rs] = pa0[rs][aoff + ws * 4];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bv[ws][ns] = pb[ns * 16 * g.POSP + ws * 4];
      }
#pragma unroll
      for (int ws = 0; ws < WS; ++ws)
#pragma unroll
        for (int rs = 0; rs < NACT; ++rs)
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns)
            acc[rs][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ws][rs], bv[ws][ns], acc[rs][ns], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NACT + NSUB, 0);
#pragma unroll
      for (int ws = 0; ws < WS; ++ws) {
        __builtin_amdgcn_sched_group_barrier(0x100, NACT + NSUB, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NACT * NSUB, 0);
      }
    }
}

// 16-byte staging of the dy tile: unit u -> (channel n, 4 consecutive positions along W)
struct DUnit { int n, pos, td, th, tw; bool valid; };
// PU = positions per unit: 4 (unit-stride dy, one float4) or 2 (dy is a stride-2 space-to-depth view: the
// float4 at the first position also holds the second one, elements 0 and 2; 1 and 3 belong to the other parity)
template <int J, int PU>
__device__ __forceinline__ DUnit dy_unit(const WgradGeom& g) {
  int u = (int)threadIdx.x + J * 256;
  asm volatile("" : "+v"(u));
  DUnit o;
  o.valid = u < g.dnunits;
  o.n = mdiv(u, g.magic_NP4);
  o.pos = PU * (u - o.n * g.np4);
  const int r1 = mdiv(o.pos, g.magic_TW);
  o.tw = o.pos - r1 * g.TW;
  o.td = mdiv(r1, g.magic_TH);
  o.th = r1 - o.td * g.TH;
  return o;
}
// position pairs load 3 dwords (elements 0 and 2 are ours): a fourth would reach past the tensor's last element
template <int PU, int NS, int J = 0, typename VT>
__device__ __forceinline__ void dy_issue_v(const WgradGeom& g, const unsigned* dchoff, const crn_rsrc& rs, int n0,
                                           int d0, int h0, int w0, VT (&val)[NS]) {
  if constexpr (J < NS) {
    unsigned goff = 0x80000000u;
    const DUnit t = dy_unit<J, PU>(g);
    if (t.valid) {
      const int od = d0 + t.td, oh = h0 + t.th, ow = w0 + t.tw;
      if (n0 + t.n < g.dy.C && od < g.dy.D && oh < g.dy.H && ow < g.dy.W)
        goff = (dchoff[t.n] + (unsigned)od * (unsigned)g.dy.sD + (unsigned)oh * (unsigned)g.dy.sH +
                (unsigned)ow * (unsigned)(PU == 2 ? 2 : 1)) * 4u;
    }
    if constexpr (PU == 4) crn_bload4(val[J], rs, goff); else crn_bload3(val[J], rs, goff);
    dy_issue_v<PU, NS, J + 1>(g, dchoff, rs, n0, d0, h0, w0, val);
  }
}
template <int PU, int NS, int J = 0, typename VT>
__device__ __forceinline__ void dy_commit_v(const WgradGeom& g, float* ldsB, const VT (&val)[NS]) {
  if constexpr (J < NS) {
    if (J * 256 < g.dnunits) {
      const DUnit t = dy_unit<J, PU>(g);
      if (t.valid) {
        float* dst = ldsB + t.n * g.POSP + t.pos;
        if constexpr (PU == 4) {
          *reinterpret_cast<f32x4*>(dst) = val[J];
        } else {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<f32x2*>(dst) = (f32x2){val[J][0], val[J][2]};
        }
      }
    }
    dy_commit_v<PU, NS, J + 1>(g, ldsB, val);
  }
}

template <int RSUB, int NSUB, bool XV, int DV>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradGeom g) {
  crn_kernarg_touch(g);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);   // x channel table; dy channel offsets at [192,256)
  float* ldsA = lds + 2 * kChTab;           // CC * PSP   (input patch)
  float* ldsB = ldsA + g.CC * g.PSP;        // NB * POSP  (dy, [n][pos])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16;
  // XCD-aware order: the channel blocks that reduce over the same tiles (same split) read the same
  // dy tiles -> keep them, and neighbouring splits, on one XCD (one L2)
  const int nbx = gridDim.x, nby = gridDim.y;
  const int lin0 = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
  const int lin = g.xcd ? xcd_remap(lin0, nbx * nby * gridDim.z) : lin0;
  const int c0 = (lin % nbx) * g.CC;
  int n0 = ((lin / nbx) % nby) * NB;
  int split = lin / (nbx * nby);
  int tps = g.tiles_per_split;
  if (g.balanced) {                        // grid.y == 1: z -> (group, n-block in group, split)
    const int z = lin / nbx;
    int gi = 0;
    while (gi + 1 < g.n_groups && z >= g.zoff[gi + 1]) ++gi;
    const int local = z - g.zoff[gi];
    const int nbi = local / g.sg[gi];
    split = local - nbi * g.sg[gi];
    n0 = (gi * g.nbpg + nbi) * NB;
    tps = g.tps[gi];
  }
  // taps that hold real weights for this block's output columns (crnTapBoxes; the full window otherwise):
  // rows = (local channel, tap inside the box)
  const TapBox tb = box_union(g.n_box, g.n_groups, g.dy.C, n0, min(n0 + NB, g.dy.C) - 1, g.kd, g.kh, g.kw);
  const int bkd = max(tb.d1 - tb.d0, 0), bkh = max(tb.h1 - tb.h0, 0), bkw = max(tb.w1 - tb.w0, 0);
  const int Tb = bkd * bkh * bkw;
  const int nrows = min(g.CC, g.x.C - c0) * Tb;
  // row sub-tile s = rs*4 + wave: the real rows are spread evenly over the four wavefronts
  const int nact = max(0, ((nrows + 15) / 16 - wave + 3) / 4);

  PatchDesc pdsc;
  pdsc.x = g.x; pdsc.tr = g.tr; pdsc.pd = g.pd; pdsc.ph = g.ph; pdsc.pw = g.pw;
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.plane = g.PH * g.PW; pdsc.lg2 = g.lg2;
  pdsc.PSP = g.PSP; pdsc.magic_PW = g.magic_PW; pdsc.magic_PD = g.magic_PD;
  pdsc.nunits = g.nunits; pdsc.plu = g.plu; pdsc.pw4 = g.pw4; pdsc.magic_PLU = g.magic_PLU; pdsc.magic_PW4 = g.magic_PW4;
  const int nplanes = g.CC * g.PD, npass = g.npass;

  // row (c_local, tap) -> LDS offset inside the patch
  int rowbase[RSUB];
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs) {
    int row = (rs * 4 + wave) * 16 + i16;
    if (row >= nrows) row = 0;                       // never stored
    const int cl = row / max(Tb, 1);
    int t = row - cl * Tb;
    const int zw = t % max(bkw, 1); t /= max(bkw, 1);
    const int zh = t % max(bkh, 1); t /= max(bkh, 1);
    rowbase[rs] = cl * g.PSP + ((t + tb.d0) * g.PH + zh + tb.h0) * g.PW + zw + tb.w0 + kk + g.lead;
  }

  f32x4 acc[RSUB][NSUB];
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[rs][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tbeg = min(split * tps, g.ntiles);
  const int tend = min(tbeg + tps, g.ntiles);
  const int npos = g.TD * g.TH * g.TW;

  float pval[XV ? 1 : PREG];
  f32x4 pv4[XV ? NVX : 1];
  unsigned inmask = 0;
  constexpr int DPU = DV == 2 ? 2 : 4, DNS = DV == 2 ? NVX : NV;     // positions per unit, float4 slots
  float dval[DV ? 8 : DREG];
  using DVT = typename std::conditional<DV == 2, f32x3, f32x4>::type;
  DVT dv4[DV ? DNS : 1];

  auto tile_origin = [&](int tl, int& b, int& d0, int& h0, int& w0) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    b = tile; d0 = tdi * g.TD; h0 = thi * g.TH; w0 = twi * g.TW;
  };
  // scalar dy staging: planes = channels nl, in-plane index r = position (td,th,tw); lanes run
  // along w: coalesced global reads, consecutive LDS addresses.  Per-thread position variants
  // (npos <= 512 -> at most 2) are hoisted per tile.
  HWVar dhw0, dhw1;
  auto dy_prepare = [&](int d0, int h0, int w0) {
    auto one = [&](int u, HWVar& out) {
      const int r = g.dlg2 >= 8 ? tid + u * 256 : (tid & ((1 << g.dlg2) - 1));
      const bool valid = r < npos && (u == 0 || g.dlg2 == 9);
      const int r1 = mdiv(r, g.magic_TW), tw = r - r1 * g.TW;
      const int td = mdiv(r1, g.magic_TH), th = r1 - td * g.TH;
      const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
      out.r = valid ? r : -1;
      out.in = valid && od < g.dy.D && oh < g.dy.H && ow < g.dy.W;
      out.off = (unsigned)od * (unsigned)g.dy.sD + (unsigned)oh * (unsigned)g.dy.sH +
                (unsigned)ow * (unsigned)g.dy.sW;
    };
    one(0, dhw0);
    one(1, dhw1);
  };
  // slot J -> (channel q, variant u); q wave-uniform when a plane spans >= 256 slots
  auto dy_issue = [&](const crn_rsrc& rs, auto jc, float& v) {
    constexpr int J = decltype(jc)::value;
    unsigned goff = 0x80000000u;
    if (g.dlg2 >= 8) {
      int q = __builtin_amdgcn_readfirstlane(g.dlg2 == 9 ? J / 2 : J);
      asm volatile("" : "+s"(q));
      const HWVar h = (J % 2 == 1 && g.dlg2 == 9) ? dhw1 : dhw0;
      if (J < g.dnpass && q < NB && n0 + q < g.dy.C)
        goff = h.in ? (choff[kChTab + q] + h.off) * 4u : 0x80000000u;
    } else {
      int jq = J * (256 >> g.dlg2);
      asm volatile("" : "+s"(jq));
      const int q = jq + (tid >> g.dlg2);
      if (J < g.dnpass && dhw0.in && q < NB && n0 + q < g.dy.C) goff = (choff[kChTab + q] + dhw0.off) * 4u;
    }
    crn_bload(v, rs, goff);
  };
  auto dy_commit = [&](auto jc, float v) {
    constexpr int J = decltype(jc)::value;
    if (g.dlg2 >= 8) {
      int q = __builtin_amdgcn_readfirstlane(g.dlg2 == 9 ? J / 2 : J);
      asm volatile("" : "+s"(q));
      const HWVar h = (J % 2 == 1 && g.dlg2 == 9) ? dhw1 : dhw0;
      if (q < NB && h.r >= 0) ldsB[q * g.POSP + h.r] = v;
    } else {
      int jq = J * (256 >> g.dlg2);
      asm volatile("" : "+s"(jq));
      const int q = jq + (tid >> g.dlg2);
      if (q < NB && dhw0.r >= 0) ldsB[q * g.POSP + dhw0.r] = v;
    }
  };
#define CRN_DY_8(OP, A, B0)                                                                        \
  if (g.dnpass > B0) {                                                                             \
    OP(A IntC<B0 + 0>{}, dval[B0 + 0]); OP(A IntC<B0 + 1>{}, dval[B0 + 1]);                        \
    OP(A IntC<B0 + 2>{}, dval[B0 + 2]); OP(A IntC<B0 + 3>{}, dval[B0 + 3]);                        \
    OP(A IntC<B0 + 4>{}, dval[B0 + 4]); OP(A IntC<B0 + 5>{}, dval[B0 + 5]);                        \
    OP(A IntC<B0 + 6>{}, dval[B0 + 6]); OP(A IntC<B0 + 7>{}, dval[B0 + 7]);                        \
  }
#define CRN_DY_8U(OP, A, B0)                                                                       \
  OP(A IntC<B0 + 0>{}, dval[B0 + 0]); OP(A IntC<B0 + 1>{}, dval[B0 + 1]);                          \
  OP(A IntC<B0 + 2>{}, dval[B0 + 2]); OP(A IntC<B0 + 3>{}, dval[B0 + 3]);                          \
  OP(A IntC<B0 + 4>{}, dval[B0 + 4]); OP(A IntC<B0 + 5>{}, dval[B0 + 5]);                          \
  OP(A IntC<B0 + 6>{}, dval[B0 + 6]); OP(A IntC<B0 + 7>{}, dval[B0 + 7]);
#define CRN_COMMA ,
  auto stage_issue = [&](int b, int d0, int h0, int w0) {
    const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
    const crn_rsrc drs = make_rsrc(g.dy.base + (int64_t)b * g.dy.sB);
    if constexpr (XV) {
      patch_issue_v<4>(pdsc, choff, xrs, c0, d0, h0, w0, pv4, inmask);
    } else {
      PatchHW phw;
      patch_prepare(pdsc, h0, w0, phw);
      patch_issue(pdsc, choff, phw, xrs, nplanes, npass, c0, d0, pval);
    }
    if constexpr (DV != 0) {
      dy_issue_v<DPU, DNS>(g, choff + kChTab, drs, n0, d0, h0, w0, dv4);
    } else {
      dy_prepare(d0, h0, w0);
      CRN_DY_8U(dy_issue, drs CRN_COMMA, 0) CRN_DY_8U(dy_issue, drs CRN_COMMA, 8)
      CRN_DY_8U(dy_issue, drs CRN_COMMA, 16) CRN_DY_8U(dy_issue, drs CRN_COMMA, 24)
    }
  };
  auto stage_commit = [&](int d0, int h0, int w0) {
    if constexpr (XV) {
      patch_commit_v<4>(pdsc, choff, ldsA, pv4, inmask);
      inmask = 0;
    } else {
      PatchHW phw;
      patch_prepare(pdsc, h0, w0, phw);
      patch_commit(pdsc, choff, phw, ldsA, nplanes, npass, c0, d0, pval);
    }
    if constexpr (DV != 0) {
      dy_commit_v<DPU, DNS>(g, ldsB, dv4);
    } else {
      dy_prepare(d0, h0, w0);
      CRN_DY_8(dy_commit, , 0) CRN_DY_8(dy_commit, , 8) CRN_DY_8(dy_commit, , 16) CRN_DY_8(dy_commit, , 24)
    }
  };

  int cb = 0, cd0 = 0, ch0 = 0, cw0 = 0;     // origin of the tile currently held in registers
  stage_choff(g.x, g.tr, choff, c0, g.CC);
  if (tid < NB) {
    const int n = min(n0 + tid, g.dy.C - 1);
    choff[kChTab + tid] = g.dy.chan_off ? (unsigned)g.dy.chan_off[n] : (unsigned)n * (unsigned)g.dy.sC;
  }
  __syncthreads();
  if (tbeg < tend) {
    tile_origin(tbeg, cb, cd0, ch0, cw0);
    stage_issue(cb, cd0, ch0, cw0);
  }
  for (int tl = tbeg; tl < tend; ++tl) {
    if constexpr (XV) crn_wait_loads4n(pv4); else crn_wait_loads(pval);
    if constexpr (DV != 0) crn_wait_loads4n(dv4); else crn_wait_loads(dval);
    __syncthreads();
    if (g.dbg != 2 || tl == tbeg) stage_commit(cd0, ch0, cw0);
    __syncthreads();
    if (tl + 1 < tend && g.dbg != 2) {
      tile_origin(tl + 1, cb, cd0, ch0, cw0);
      stage_issue(cb, cd0, ch0, cw0);
    }
    // reduction over the tile's positions (wgrad_rows; the switches stay outside the loops)
    if (g.dbg != 1) {
      const int bbase = i16 * g.POSP + kk;
#define CRN_WG_WS(NACT)                                                                           \
  switch (g.TW >> 2) {                                                                            \
    case 1: wgrad_rows<1, RSUB, NSUB, NACT>(acc, ldsA, ldsB, rowbase, bbase, g); break;           \
    case 2: wgrad_rows<2, RSUB, NSUB, NACT>(acc, ldsA, ldsB, rowbase, bbase, g); break;           \
    case 3: wgrad_rows<3, RSUB, NSUB, NACT>(acc, ldsA, ldsB, rowbase, bbase, g); break;           \
    default: wgrad_rows<4, RSUB, NSUB, NACT>(acc, ldsA, ldsB, rowbase, bbase, g); break;          \
  }
      if constexpr (RSUB >= 4) {
        if (nact * 8 <= RSUB * 4) { CRN_WG_WS((RSUB * 4) / 8) }
        else if (nact * 8 <= RSUB * 5) { CRN_WG_WS((RSUB * 5 + 7) / 8) }
        else if (nact * 8 <= RSUB * 6) { CRN_WG_WS((RSUB * 6) / 8) }
        else { CRN_WG_WS(RSUB) }
      } else {
        CRN_WG_WS(RSUB)
      }
#undef CRN_WG_WS
    }
  }
#undef CRN_DY_8
#undef CRN_DY_8U
#undef CRN_COMMA

  // D row = kk*4 + r -> weight row (c_local, tap); col = i16 -> n
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row_ = (rs * 4 + wave) * 16 + kk * 4 + r;
      if (row_ >= nrows) continue;
      const int cl = row_ / max(Tb, 1);
      int tl = row_ - cl * Tb;
      const int zw = tl % max(bkw, 1); tl /= max(bkw, 1);
      const int zh = tl % max(bkh, 1); tl /= max(bkh, 1);
      const int64_t prow = (int64_t)(c0 + cl) * g.Tfull +
                           ((tl + tb.d0 + g.bd0) * g.khf + zh + tb.h0 + g.bh0) * g.kwf + zw + tb.w0 + g.bw0;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const int n = n0 + ns * 16 + i16;
        if (n < g.ncols)
          atomicAdd(g.dw + prow * g.Npad + n, acc[rs][ns][r]);
      }
    }
}


template <int MSUB, int NSUB, int XV>
inline int launch_fwd(const ConvGeom& g, dim3 grid, size_t lds_bytes, hipStream_t st) {
  auto k = conv_fwd_kernel<MSUB, NSUB, XV>;
  if (lds_bytes > 65536)
    CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

template <int RSUB, int NSUB, bool XV, int DV>
inline int launch_wgrad(const WgradGeom& g, dim3 grid, size_t lds_bytes, hipStream_t st) {
  auto k = conv_wgrad_kernel<RSUB, NSUB, XV, DV>;
  if (lds_bytes > 65536)
    CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

}  // namespace crnk

// split-K scratch + reduction shared by the fp32 and the bf16x3 engine (defined in conv_igemm.hip)
float* crn_splitk_scratch(size_t floats, hipStream_t st);      // per (device, stream)
int* crn_splitk_counters(size_t n);          // n zero-initialised arrival counters (self-resetting), or nullptr
int crn_splitk_reduce(const crnView& y, const float* scratch, int splits, int accumulate, hipStream_t st);

// launchers defined in conv_inst.hip (one object per configuration)
#define CRN_FWD_CONFIGS(X) X(8, 1) X(4, 2) X(4, 1) X(2, 4) X(2, 2) X(2, 1) X(1, 4) X(1, 2) X(1, 1)
#define CRN_WG_CONFIGS(X) X(8, 1) X(4, 2) X(4, 1) X(2, 4) X(2, 2) X(2, 1) X(1, 4) X(1, 2) X(1, 1)
#define CRN_DECL_FWD(M, N) int crn_launch_fwd_##M##_##N(const crnk::ConvGeom&, int xvec, dim3, size_t, hipStream_t);
#define CRN_DECL_WG(R, N) int crn_launch_wgrad_##R##_##N(const crnk::WgradGeom&, int xvec, int dyvec /* 0 1 2 */, dim3, size_t, hipStream_t);
CRN_FWD_CONFIGS(CRN_DECL_FWD)
CRN_WG_CONFIGS(CRN_DECL_WG)
