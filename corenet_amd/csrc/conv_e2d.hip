// Split-bf16 ("bf16x3") MFMA engine for the encoder's small convolutions (gfx950 / CDNA4).
//
// The ResNet-50 encoder (resnet50.py:49-115) at batch 4 is 52 convolutions of 0.5-1.2 GFLOP each: 1x1 and 3x3
// windows over 64^2 ... 8^2 images with 64 ... 2048 channels.  None of them is bound by arithmetic (1.2 GFLOP is
// 8 us of fp32 MFMA on the whole chip); what the general engines (conv_igemm.hip / conv_bf3.hip) spend on them is
// a chain of load -> barrier -> MFMA steps, 14 us per 1x1 and 30 + 5 us (kernel + split-K reduction) per 3x3.
// This engine is built around that chain instead of around MFMA throughput:
//
//  * weights arrive ALREADY in MFMA operand order and already split into bf16 hi / lo terms (crn_bf3_operands,
//    one launch per weight pack for all layers): operand block (K step, 16 output columns) = 64 lanes x 32 bytes,
//    so a wave reads the B operand of a K step with two coalesced 16-byte buffer loads straight into the
//    registers the MFMA reads.  Weights never touch LDS, and there is no barrier and no VALU work on their path.
//  * a workgroup = 4 waves = 64 output positions x 64 output columns (wave w owns columns 16w..16w+15 for all 64
//    positions: 4 accumulator tiles); 256 ... 1024 workgroups per layer, split-K over input channels when the
//    tiles alone do not fill the 256 CUs (partial sums to the shared scratch + crn_splitk_reduce).
//  * a chunk = KS K steps of 32 (3x3: 32 channels x 9 taps; 1x1: 128 channels).  The B registers of a whole
//    chunk are resident; the registers of K step s are re-loaded with the next chunk's step s right after the
//    MFMAs that consumed them, and the next chunk's input patch is loaded before the MFMAs of this chunk start,
//    so every load has one chunk of MFMAs (0.3-0.8 us) to land and a chunk costs ONE barrier.  Outstanding loads
//    are tracked by hand (s_waitcnt vmcnt(N) with the N the issue order implies): the compiler's own waitcnt
//    pass would serialise them.
//  * the input patch is staged like in conv_bf3.hip: fp32 from HBM, BatchRenorm-apply + ReLU of the producer
//    fused, split into bf16 hi / lo, stored as [position][32 channels] rows (one ds_read_b128 per operand).
//    K of v_mfma_f32_16x16x32_bf16 = 32 input channels of ONE window tap, so a tap is an LDS row offset.
//
// Same operation as crn_conv_fwd (y = bias + window correlation of T(x) with packed weights); serves the forward
// pass and the data gradient (packed data-gradient weights).  Numerics: hi*hi + hi*lo + lo*hi with fp32
// accumulation, ~2^-16 relative per product, as conv_bf3.hip.
#include "conv_kernels.h"
#include <algorithm>
#include <cstdlib>

namespace {
using namespace crnk;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct E2dGeom {
  crnView x, y;
  crnInTransform tr;
  const void* wop;        // operand blocks [K step][n tile][64 lanes][hi 16 B | lo 16 B]
  const float* bias;
  int bias_sB;
  int ntn;                // n tiles of 16 columns (Npad / 16)
  int nblk;               // 32-channel blocks of the input (Cin / 32)
  int ph, pw;             // window origin (3x3)
  int tilesH, tilesW;     // 3x3: tiles per image; 1x1: tilesW = tiles per sample, tilesH = 1
  int nchunks, chunks_per_split;
  int tab;                // entries of the scale / shift tables in LDS (channels of one split)
  int mode;               // 0 store, 1 accumulate (read-modify-write), 3 split-K partial sums [split][b][n][pos]
  int dbg;                // tuning aid (CRN_E2D_DBG): 1 return at once, 2 no chunk loop, 4 no MFMAs, 8 no stores,
                          // 16 shader-clock stamps of workgroup 0 / wave 0 into stamps[] (crn_e2d_debug_stamps)
  long long* stamps;
};

// crn_bf3_operands: one row of the layer table = (first float of the packed [Cin][T][Npad] weights, first 32-byte
// entry of the layer in the output, Cin, T, Npad, first workgroup of the layer, KHW: 0 = MFMA operand blocks of the
// encoder engine below; kh*kw > 0 = slab order of conv_bf3.hip, [chunk of 8 channels][zd][tap slot][n])
constexpr int kOpFields = 7;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// packed fp32 weights -> operand blocks.  entry (kg, ntile, lane = kk*16 + i16): the 8 channels
// c = 32*(kg / T) + 8*kk + j of tap kg % T for output column 16*ntile + i16, as 8 hi and 8 lo bf16.
__global__ __launch_bounds__(256) void bf3_operands_kernel(const float* packed, const long long* desc, int nlayers,
                                                           char* out) {
  int lo = 0, hi = nlayers - 1;
  while (lo < hi) {                                   // last layer whose first workgroup is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * kOpFields + 5] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + lo * kOpFields;
  const int Cin = (int)d[2], T = (int)d[3], Npad = (int)d[4];
  if (d[6] > 0) {                                     // slab order: entry ((chunk*kd + zd)*TP + tp)*Npad + n
    const int KHW = (int)d[6], kd = T / KHW, TP = (KHW + 3) & ~3;
    const long long entries = (long long)((Cin + 7) >> 3) * kd * TP * Npad;
    const long long e = ((long long)blockIdx.x - d[5]) * 256 + threadIdx.x;
    if (e >= entries) return;
    const int n = (int)(e % Npad);
    long long r = e / Npad;
    const int tp = (int)(r % TP); r /= TP;
    const int zd = (int)(r % kd);
    const int chunk = (int)(r / kd);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = chunk * 8 + j;
      v[j] = (c < Cin && tp < KHW) ? packed[d[0] + ((long long)c * T + zd * KHW + tp) * Npad + n] : 0.f;
    }
    bf16x8 h, l;
    split8(v, h, l);
    bf16x8* dst = reinterpret_cast<bf16x8*>(out + (d[1] + e) * 32);
    dst[0] = h;
    dst[1] = l;
    return;
  }
  const int ntn = Npad >> 4;
  const long long entries = (long long)(Cin >> 5) * T * ntn * 64;
  const long long e = ((long long)blockIdx.x - d[5]) * 256 + threadIdx.x;
  if (e >= entries) return;
  const int lane = (int)(e & 63);
  const long long rest = e >> 6;
  const int ntile = (int)(rest % ntn);
  const int kg = (int)(rest / ntn);
  const int cbg = kg / T, t = kg - cbg * T;
  const int kk = lane >> 4, i16 = lane & 15;
  const float* src = packed + d[0] + ((long long)(cbg * 32 + kk * 8) * T + t) * Npad + ntile * 16 + i16;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = src[(long long)j * T * Npad];
  bf16x8 h, l;
  split8(v, h, l);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out + (d[1] + e) * 32);
  dst[0] = h;
  dst[1] = l;
}

constexpr unsigned kOOB = 0x80000000u;     // buffer offset outside the 2 GiB range of make_rsrc: the load returns 0

template <int N>
__device__ __forceinline__ void wait_vm(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// T = window taps (9: 3x3, 1: 1x1); CB = 32-channel blocks per chunk; TW = tile width (3x3: 16 or 8, tile = 64/TW
// rows; 1x1: the 64 positions of a tile are consecutive in the flattened image and TW only names the instance)
template <int T, int CB, int TW>
__global__ __launch_bounds__(256, 2) void conv_e2d_kernel(E2dGeom g) {
  constexpr int KW = T == 9 ? 3 : 1;
  constexpr int TH = 64 / TW, PH = TH + KW - 1, PW = TW + KW - 1, NPOS = PH * PW;
  constexpr int NPA = T == 9 ? 128 : 64;               // position rows per channel block in LDS
  constexpr int KS = CB * T;                           // K steps per chunk
  constexpr int NPL = T == 9 ? 16 : 8;                 // patch load instructions per thread and chunk
  // LDS layout of a patch plane (hi or lo): [channel block][kk = 8-channel group][position], 16-byte units; the
  // 16 lanes of an MFMA row group read 16 consecutive units.  PS = NPA + 4: consecutive kk planes start 16 banks
  // apart, so the staging writes (4 kk x 16 positions per wave instruction) are conflict free as well
  constexpr int PS = NPA + 4;
  constexpr int PLANE = CB * 4 * PS;                   // units of one plane of one buffer
  static_assert(T == 9 ? CB == 1 : CB == 4, "thread -> patch element mapping");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tscale = reinterpret_cast<float*>(smem);
  float* tshift = tscale + g.tab;
  bf16x8* Abuf = reinterpret_cast<bf16x8*>(smem + (size_t)g.tab * 8);      // [2 buffers][hi, lo][PLANE]

  const long long t_entry = (long long)__builtin_amdgcn_s_memtime();
  crn_kernargs_now(g.x.base, g.x.B, g.x.C, g.x.H, g.x.W, g.x.sB, g.x.sC, g.x.sH, g.y.base, g.y.C, g.y.sB, g.y.sC, g.y.sH,
               g.tr.scale, g.tr.shift, g.tr.pre_relu, g.tr.post_relu, g.wop, g.bias, g.bias_sB, g.ntn, g.nblk, g.ph,
               g.pw, g.tilesH, g.tilesW, g.nchunks, g.chunks_per_split, g.tab, g.mode, g.dbg, g.stamps, gridDim.x);
  if (g.dbg & 1) return;
  const bool stamp = (g.dbg & 16) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  auto mark = [&](int i) { if (stamp) g.stamps[i] = (long long)__builtin_amdgcn_s_memtime() - t_entry; };
  mark(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int b = tile;
  const int split = blockIdx.z;
  const int cbeg = split * g.chunks_per_split, cend = min(cbeg + g.chunks_per_split, g.nchunks);
  const int ch0 = cbeg * CB * 32;                       // first channel of this split (origin of the tables)
  const bool has_tr = g.tr.scale != nullptr;

  const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
  const crn_rsrc wrs = make_rsrc(reinterpret_cast<const float*>(g.wop));

  // ---- patch staging geometry ----
  // 3x3: thread = (kk8 = tid & 3: channels 8*kk8 .. +7 of the chunk, patch position q = (tid >> 2) + 64*pass)
  // 1x1: thread = (kk8 = tid & 3, position quad pq = (tid >> 2) & 15, channel block cbl = wave); the four
  //      positions 4*pq + i of a quad go to LDS rows i*16 + pq, so that M tile ms = rows 16*ms .. +15 holds the
  //      positions 4*i16 + ms and both the staging writes and the operand reads are contiguous per instruction
  const int kk8 = tid & 3;
  unsigned poff[2];                                    // byte offset of the thread's position(s) in the sample
  bool pin[2] = {false, false};
  int h0 = 0, w0 = 0, p0 = 0;
  if constexpr (T == 9) {
    h0 = thi * TH; w0 = twi * TW;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int q = (tid >> 2) + 64 * p;
      const int r = q / PW, c = q - r * PW;
      const int gh = h0 - g.ph + r, gw = w0 - g.pw + c;
      pin[p] = q < NPOS && gh >= 0 && gh < g.x.H && gw >= 0 && gw < g.x.W;
      poff[p] = pin[p] ? (unsigned)(gh * g.x.sH + gw) * 4u : kOOB;
    }
  } else {
    p0 = twi * 64;
    poff[0] = (unsigned)(p0 + 4 * ((tid >> 2) & 15)) * 4u;
    poff[1] = 0;
  }
  const unsigned sC4 = (unsigned)g.x.sC * 4u;

  float pf[T == 9 ? 16 : 1];                           // 3x3: 2 positions x 8 channels, one dword load each
  f32x4 pv[T == 9 ? 1 : 8];                            // 1x1: 8 channels x 4 consecutive positions, 16-byte loads
  auto patch_issue = [&](int chunk) {
    if constexpr (T == 9) {
      const unsigned oob = chunk < cend ? 0u : kOOB;   // poff already carries kOOB for padding positions
      const unsigned cbase = (unsigned)(chunk * 32 + kk8 * 8) * sC4;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) crn_bload(pf[p * 8 + j], xrs, (poff[p] + cbase + (unsigned)j * sC4) | oob);
    } else {
      const int cbg = chunk * CB + wave;
      const unsigned oob = (chunk < cend && cbg < g.nblk) ? 0u : kOOB;
      const unsigned cbase = (unsigned)(cbg * 32 + kk8 * 8) * sC4;
#pragma unroll
      for (int j = 0; j < 8; ++j) crn_bload4(pv[j], xrs, (poff[0] + cbase + (unsigned)j * sC4) | oob);
    }
  };
  auto transform8 = [&](float (&v)[8], int ctab) {     // ctab = table index of the first of the 8 channels
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = v[j];
      if (g.tr.pre_relu) a = fmaxf(a, 0.f);
      a = a * tscale[ctab + j] + tshift[ctab + j];
      if (g.tr.post_relu) a = fmaxf(a, 0.f);
      v[j] = a;
    }
  };
  auto patch_commit = [&](int chunk, bf16x8* Ahi, bf16x8* Alo) {
    if constexpr (T == 9) {
      const int ctab = (chunk - cbeg) * 32 + kk8 * 8;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pf[p * 8 + j];
        if (has_tr && pin[p]) transform8(v, ctab);     // zero padding stays zero
        bf16x8 h, l;
        split8(v, h, l);
        const int q = (tid >> 2) + 64 * p;
        Ahi[kk8 * PS + q] = h;
        Alo[kk8 * PS + q] = l;
      }
    } else {
      const int cbg = chunk * CB + wave;
      const int ctab = (chunk - cbeg) * CB * 32 + wave * 32 + kk8 * 8;
      const int pq = (tid >> 2) & 15;
      const bool live = cbg < g.nblk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pv[j][i];
        if (has_tr && live) transform8(v, ctab);
        bf16x8 h, l;
        split8(v, h, l);
        const int row = i * 16 + pq;
        Ahi[(wave * 4 + kk8) * PS + row] = h;
        Alo[(wave * 4 + kk8) * PS + row] = l;
      }
    }
  };
  auto patch_wait = [&]() {                            // the B loads of the next chunk (2 per K step) were issued later
    if constexpr (T == 9) {
#pragma unroll
      for (int i = 0; i < 16; i += 8)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(pf[i]), "+v"(pf[i + 1]), "+v"(pf[i + 2]), "+v"(pf[i + 3]), "+v"(pf[i + 4]), "+v"(pf[i + 5]),
                       "+v"(pf[i + 6]), "+v"(pf[i + 7])
                     : "n"(2 * KS));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pv[i]) : "n"(2 * KS));
    }
  };

  // ---- weight operands: registers of a whole chunk ----
  f32x4 bh[KS], bl[KS];
  const int ntile = blockIdx.y * 4 + wave;
  const unsigned lb = (unsigned)(ntile * 64 + lane) * 32u;
  const unsigned kstride = (unsigned)g.ntn * 2048u;    // bytes between consecutive K steps
  // oob = 0, or kOOB for a K step that does not exist (offsets stay below 2^31, so OR-ing the top bit moves the
  // load out of the buffer's range: zeros, no memory traffic, no branch)
  auto b_issue = [&](int chunk, int s, unsigned oob, f32x4& h, f32x4& l) {
    const unsigned off = (lb + (unsigned)(chunk * KS + s) * kstride) | oob;
    crn_bload4(h, wrs, off);
    crn_bload4(l, wrs, off + 16u);
  };
  auto step_oob = [&](int chunk, int s) -> unsigned {   // wave-uniform
    return (chunk < cend && chunk * CB + s / T < g.nblk) ? 0u : kOOB;
  };
  // operand read offsets (bf16x8 units) of the lane's 4 M tiles at tap (0, 0)
  int aoff[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    if constexpr (T == 9) aoff[ms] = kk * PS + (ms * (16 / TW) + i16 / TW) * PW + (i16 % TW);
    else aoff[ms] = kk * PS + 16 * ms + i16;
  }

  // two accumulator sets: hi*hi, and the two small cross terms (12 MFMAs per K step, consecutive ones independent)
  f32x4 acc[4], acx[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) { acc[ms] = (f32x4){0.f, 0.f, 0.f, 0.f}; acx[ms] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  if (cbeg < cend) {
    // prologue: everything of the first chunk in flight at once, one wait
#pragma unroll
    for (int s = 0; s < KS; ++s) b_issue(cbeg, s, step_oob(cbeg, s), bh[s], bl[s]);
    patch_issue(cbeg);
    mark(1);
    // scale / shift of this split's channels (compiler-visible loads, issued behind the hand-tracked ones: the
    // compiler's wait for them is the prologue's one round trip to memory)
    if (has_tr)
      for (int c = tid; c < g.tab; c += 256) {
        const int cc = min(ch0 + c, g.x.C - 1);
        tscale[c] = g.tr.scale[cc];
        tshift[c] = g.tr.shift[cc];
      }
    if constexpr (T == 9) {
#pragma unroll
      for (int i = 0; i < 16; i += 8)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(pf[i]), "+v"(pf[i + 1]), "+v"(pf[i + 2]), "+v"(pf[i + 3]), "+v"(pf[i + 4]), "+v"(pf[i + 5]),
                       "+v"(pf[i + 6]), "+v"(pf[i + 7]));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[i]));
    }
    mark(2);
    __syncthreads();                                   // the tables
    patch_commit(cbeg, Abuf, Abuf + PLANE);
    __syncthreads();
    mark(3);

    for (int chunk = cbeg; chunk < ((g.dbg & 2) ? cbeg : cend); ++chunk) {
      const int buf = (chunk - cbeg) & 1;
      const bf16x8* Ahi = Abuf + buf * 2 * PLANE;
      const bf16x8* Alo = Ahi + PLANE;
      patch_issue(chunk + 1);                          // out-of-range (zeros, no traffic) after the last chunk
      // the A operands of K step s + 1 are read from LDS before the MFMAs of step s are issued (a workgroup may be
      // alone on its CU, one wave per SIMD: nothing else hides the LDS latency)
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) { ah[ms] = Ahi[aoff[ms]]; al[ms] = Alo[aoff[ms]]; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // loads issued after this step's: the rest of this chunk's steps, the next patch, the next chunk's steps < s
        wait_vm<2 * (KS - 1) + NPL>(bh[s], bl[s]);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[s]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, bl[s]);
        bf16x8 nh[4], nl[4];
        if (s + 1 < KS && !(g.dbg & 32)) {
          const int cbl = (s + 1) / T, t = (s + 1) % T;
          const int toff = cbl * 4 * PS + (t / KW) * PW + (t % KW);
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) { nh[ms] = Ahi[aoff[ms] + toff]; nl[ms] = Alo[aoff[ms] + toff]; }
        }
        __builtin_amdgcn_sched_barrier(0);             // (the scheduler would sink the reads below the MFMAs)
        // (1x1: channel blocks past Cin are zeros on both sides -- out-of-range loads -- and are multiplied anyway)
        if (!(g.dbg & 4)) {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) acc[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ms], wh, acc[ms], 0, 0, 0);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) acx[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ms], wl, acx[ms], 0, 0, 0);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) acx[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ms], wh, acx[ms], 0, 0, 0);
        }
        b_issue(chunk + 1, s, (g.dbg & 64) ? kOOB : step_oob(chunk + 1, s), bh[s], bl[s]);
        if (s + 1 < KS && !(g.dbg & 32)) {
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) { ah[ms] = nh[ms]; al[ms] = nl[ms]; }
        }
      }
      mark(4 + 2 * (chunk - cbeg));
      patch_wait();
      mark(5 + 2 * (chunk - cbeg));
      if (chunk + 1 < cend) {
        bf16x8* Nhi = Abuf + (buf ^ 1) * 2 * PLANE;
        patch_commit(chunk + 1, Nhi, Nhi + PLANE);
      }
      __syncthreads();
    }
  }

  // the loads issued for "the chunk after the last" (out of range: zeros) still target these registers: keep them
  // allocated until the loads have landed
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bh[s]), "+v"(bl[s]));
  if constexpr (T == 9) {
#pragma unroll
    for (int i = 0; i < 16; i += 8)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(pf[i]), "+v"(pf[i + 1]), "+v"(pf[i + 2]), "+v"(pf[i + 3]), "+v"(pf[i + 4]), "+v"(pf[i + 5]),
                     "+v"(pf[i + 6]), "+v"(pf[i + 7]));
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[i]));
  }

  mark(30);
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) acc[ms] += acx[ms];
  // ---- epilogue: D row = 4*kk + r (position of the M tile), col = i16 (output column) ----
  const int n = ntile * 16 + i16;
  if (n >= g.y.C || (g.dbg & 8)) return;
  const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
  float* yb = g.y.base + (int64_t)(g.mode == 3 ? split * g.x.B + b : b) * g.y.sB + (int64_t)n * g.y.sC;
  if constexpr (T == 9) {
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int rr = 4 * kk;                            // first of the lane's 4 positions inside the M tile
      const int oh = h0 + ms * (16 / TW) + rr / TW, ow = w0 + rr % TW;
      float* dst = yb + (int64_t)oh * g.y.sH + ow;
      f32x4 v = acc[ms] + bsv;
      if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = yb + p0 + 16 * kk + 4 * r;           // positions 4*(4*kk + r) + ms, ms = 0..3
      f32x4 v = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]} + bsv;
      if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
  mark(31);
}

// ---- weight gradient of the 1x1 layers ---------------------------------------------------------------------------
// dw[c][n] += sum over (b, position) of T(x)[b, c, position] * dy[b, n, position]: a [C x positions] . [positions x N]
// GEMM whose K = positions is the contiguous index of BOTH operands in memory, and an MFMA operand register holds 8
// consecutive K of one row: lane (row i16, K slice kk) loads x[b][c0 + i16][s + 8*kk .. +7] (and the same of dy) with two
// 16-byte loads, splits them to bf16 hi / lo in registers and multiplies -- no LDS on the operand path, no barrier.
// Workgroup = 4 waves on one 64 (c) x 32 (n) tile, each wave every fourth K step of the workgroup's K range (the 36
// launches per step have K = 256 ... 16384 positions); the four partial tiles meet in LDS and go to dw with
// fire-and-forget atomics (dw is zeroed by the caller or by zero_first, like crn_conv_wgrad).
struct Wg1Geom {
  const float* x; const float* dy; float* dw;
  const float* scale; const float* shift;
  int pre_relu, post_relu;
  long long xsB, dsB;      // sample strides
  int C, N, Npad, S;       // S = positions per sample (multiple of 32)
  int ksteps, ksteps_per_block;   // K steps of 32 positions: B * S / 32 in total
};

__global__ __launch_bounds__(256, 2) void wgrad1x1_bf3_kernel(Wg1Geom g) {
  crn_kernargs_now(g.x, g.dy, g.dw, g.scale, g.shift, g.pre_relu, g.post_relu, g.xsB, g.dsB, g.C, g.N, g.Npad, g.S,
                   g.ksteps, g.ksteps_per_block);
  __shared__ __attribute__((aligned(16))) f32x4 red[4][8][64];          // [wave][tile][lane]: 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // scalar: the buffer descriptors depend on it
  const int i16 = lane & 15, kk = lane >> 4;
  const int c0 = blockIdx.x * 64, n0 = blockIdx.y * 32;
  const int kbeg = blockIdx.z * g.ksteps_per_block, kend = min(kbeg + g.ksteps_per_block, g.ksteps);
  // rows of this lane: 4 x channels (A), 2 x output columns (B); rows past the tensors are loaded as zeros
  unsigned arow[4], brow[2];
  float sc[4], sh[4];
  const bool has_tr = g.scale != nullptr;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int c = c0 + 16 * mt + i16;
    arow[mt] = c < g.C ? (unsigned)c * (unsigned)g.S * 4u : kOOB;
    sc[mt] = (has_tr && c < g.C) ? g.scale[c] : 1.f;
    sh[mt] = (has_tr && c < g.C) ? g.shift[c] : 0.f;
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + 16 * nt + i16;
    brow[nt] = n < g.N ? (unsigned)n * (unsigned)g.S * 4u : kOOB;
  }
  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[4][2], rb[2][2];                             // raw fp32 operands of one K step
  auto issue = [&](int step) {
    const bool live = step < kend;
    const int pos = step * 32, b = live ? pos / g.S : 0, s = pos - b * g.S;
    const crn_rsrc xrs = make_rsrc(g.x + (long long)b * g.xsB);
    const crn_rsrc drs = make_rsrc(g.dy + (long long)b * g.dsB);
    const unsigned koff = (unsigned)(s + 8 * kk) * 4u, oob = live ? 0u : kOOB;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      crn_bload4(ra[mt][0], xrs, (arow[mt] + koff) | oob);
      crn_bload4(ra[mt][1], xrs, (arow[mt] + koff + 16u) | oob);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      crn_bload4(rb[nt][0], drs, (brow[nt] + koff) | oob);
      crn_bload4(rb[nt][1], drs, (brow[nt] + koff + 16u) | oob);
    }
  };
  auto wait_all = [&]() {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[mt][0]), "+v"(ra[mt][1]));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[nt][0]), "+v"(rb[nt][1]));
  };
  int step = kbeg + wave;
  issue(step);
  for (; step < kend; step += 4) {
    wait_all();
    bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = ra[mt][j >> 2][j & 3];
        if (has_tr) {
          if (g.pre_relu) a = fmaxf(a, 0.f);
          a = a * sc[mt] + sh[mt];
          if (g.post_relu) a = fmaxf(a, 0.f);
        }
        v[j] = a;
      }
      if (has_tr && arow[mt] == kOOB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;           // channels past C: T(0) is not 0
      }
      split8(v, ah[mt], al[mt]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rb[nt][j >> 2][j & 3];
      split8(v, bh[nt], bl[nt]);
    }
    issue(step + 4);                                    // the raw registers are free again: next K step of this wave
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    }
  }
  wait_all();                                           // the loads issued for "the step after the last" (zeros)
  // the four partial tiles -> LDS -> one sum per element -> dw.  D row = 4*kk + r (channel), col = i16 (column)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) red[wave][mt * 2 + nt][lane] = acc[mt][nt];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave * 2 + q;                      // wave w sums tiles 2w, 2w + 1
    const f32x4 v = red[0][tile][lane] + red[1][tile][lane] + red[2][tile][lane] + red[3][tile][lane];
    const int mt = tile >> 1, nt = tile & 1;
    const int n = n0 + 16 * nt + i16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + 16 * mt + 4 * kk + r;
      if (c < g.C && n < g.N) atomicAdd(g.dw + (long long)c * g.Npad + n, v[r]);
    }
  }
}

// ---- weight gradient of the 3x3 layers: the same idea -----------------------------------------------------------
// Row m = c*9 + zh*3 + zw of dw is x shifted by (zh - 1, zw - 1).  The three zw taps of one (c, zh) read the SAME
// 8-aligned group of positions (s .. s+7 of image row h + zh - 1; W is a multiple of 8, so a group never leaves its
// row) shifted by -1 / 0 / +1: a lane owns one q = 3*c + zh, loads the aligned group (two 16-byte loads) plus the two
// edge elements (two dword loads, out of range at the image border), and builds three operand vectors from the same
// registers -- rows 3*q + 0, 1, 2, i.e. lane i16 of three M tiles.  Nothing is loaded unaligned, nothing outside
// the sample is touched, zero padding is a load that is not issued.
struct Wg3Geom {
  const float* x; const float* dy; float* dw;
  const float* scale; const float* shift;
  int pre_relu, post_relu;
  long long xsB, dsB;
  int C, N, Npad, S, H, W;
  int ksteps, ksteps_per_block;
};

__global__ __launch_bounds__(256, 2) void wgrad3x3_bf3_kernel(Wg3Geom g) {
  crn_kernargs_now(g.x, g.dy, g.dw, g.scale, g.shift, g.pre_relu, g.post_relu, g.xsB, g.dsB, g.C, g.N, g.Npad, g.S, g.H,
                   g.W, g.ksteps, g.ksteps_per_block);
  __shared__ __attribute__((aligned(16))) f32x4 red[4][12][64];         // [wave][tile][lane]: 48 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kk = lane >> 4;
  const int q0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int kbeg = blockIdx.z * g.ksteps_per_block, kend = min(kbeg + g.ksteps_per_block, g.ksteps);
  const bool has_tr = g.scale != nullptr;
  // the lane's two (c, zh) pairs and two output columns
  unsigned arow[2], brow[2];
  int dzh[2];
  float sc[2], sh[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 16 * qt + i16, c = q / 3;
    dzh[qt] = q - 3 * c - 1;                            // zh - 1
    arow[qt] = c < g.C ? (unsigned)c * (unsigned)g.S * 4u : kOOB;
    sc[qt] = (has_tr && c < g.C) ? g.scale[c] : 1.f;
    sh[qt] = (has_tr && c < g.C) ? g.shift[c] : 0.f;
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + 16 * nt + i16;
    brow[nt] = n < g.N ? (unsigned)n * (unsigned)g.S * 4u : kOOB;
  }
  f32x4 acc[6][2];
#pragma unroll
  for (int mt = 0; mt < 6; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[2][2], rb[2][2];
  float el[2], er[2];                                   // edge elements left / right of the aligned group
  unsigned vmask = 0;                                   // bits qt: group valid, 2 + qt: left edge valid, 4 + qt: right
  auto issue = [&](int step) {
    const bool live = step < kend;
    const int pos = step * 32, b = live ? pos / g.S : 0, s = pos - b * g.S + 8 * kk;
    const int h = s / g.W, w0 = s - h * g.W;
    const crn_rsrc xrs = make_rsrc(g.x + (long long)b * g.xsB);
    const crn_rsrc drs = make_rsrc(g.dy + (long long)b * g.dsB);
    vmask = 0;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int hh = h + dzh[qt];
      const bool ok = live && arow[qt] != kOOB && hh >= 0 && hh < g.H;
      const bool okl = ok && w0 > 0, okr = ok && w0 + 8 < g.W;
      const unsigned off = arow[qt] + (unsigned)(s + dzh[qt] * g.W) * 4u;
      vmask |= (ok ? 1u : 0u) << qt | (okl ? 1u : 0u) << (2 + qt) | (okr ? 1u : 0u) << (4 + qt);
      crn_bload4(ra[qt][0], xrs, ok ? off : kOOB);
      crn_bload4(ra[qt][1], xrs, ok ? off + 16u : kOOB);
      crn_bload(el[qt], xrs, okl ? off - 4u : kOOB);
      crn_bload(er[qt], xrs, okr ? off + 32u : kOOB);
    }
    const unsigned koff = (unsigned)s * 4u, oob = live ? 0u : kOOB;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      crn_bload4(rb[nt][0], drs, (brow[nt] + koff) | oob);
      crn_bload4(rb[nt][1], drs, (brow[nt] + koff + 16u) | oob);
    }
  };
  auto wait_all = [&]() {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[qt][0]), "+v"(ra[qt][1]), "+v"(el[qt]), "+v"(er[qt]));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[nt][0]), "+v"(rb[nt][1]));
  };
  auto tf = [&](float a, int qt, bool valid) -> float {  // transform of a loaded element; padding stays zero
    if (!has_tr) return a;
    if (g.pre_relu) a = fmaxf(a, 0.f);
    a = a * sc[qt] + sh[qt];
    if (g.post_relu) a = fmaxf(a, 0.f);
    return valid ? a : 0.f;
  };
  int step = kbeg + wave;
  issue(step);
  for (; step < kend; step += 4) {
    wait_all();
    bf16x8 ah[6], al[6], bh[2], bl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float u[10];                                      // positions s-1 .. s+8 of the lane's row
      u[0] = tf(el[qt], qt, (vmask >> (2 + qt)) & 1u);
#pragma unroll
      for (int j = 0; j < 8; ++j) u[1 + j] = tf(ra[qt][j >> 2][j & 3], qt, (vmask >> qt) & 1u);
      u[9] = tf(er[qt], qt, (vmask >> (4 + qt)) & 1u);
#pragma unroll
      for (int zw = 0; zw < 3; ++zw) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = u[j + zw];
        split8(v, ah[qt * 3 + zw], al[qt * 3 + zw]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rb[nt][j >> 2][j & 3];
      split8(v, bh[nt], bl[nt]);
    }
    issue(step + 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 6; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 6; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 6; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    }
  }
  wait_all();
#pragma unroll
  for (int mt = 0; mt < 6; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) red[wave][mt * 2 + nt][lane] = acc[mt][nt];
  __syncthreads();
  // D row = 4*kk + r = lane-row index iq of the M tile (qt, zw): q = q0 + 16*qt + iq, dw row 3*q + zw
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tile = wave * 3 + j;                      // wave w sums tiles 3w .. 3w+2
    const f32x4 v = red[0][tile][lane] + red[1][tile][lane] + red[2][tile][lane] + red[3][tile][lane];
    const int mt = tile >> 1, nt = tile & 1, qt = mt / 3, zw = mt - 3 * qt;
    const int n = n0 + 16 * nt + i16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 16 * qt + 4 * kk + r;
      if (q < 3 * g.C && n < g.N) atomicAdd(g.dw + (long long)(3 * q + zw) * g.Npad + n, v[r]);
    }
  }
}

bool dense_2d(const crnView& v) {   // unit W stride, rows back to back, 16-byte aligned channel planes
  return v.chan_off == nullptr && v.sW == 1 && v.sH == v.W && (v.D == 1 || v.sD == v.H * v.W) && (v.sC & 3) == 0 &&
         (v.sB & 3) == 0 && (((uintptr_t)v.base) & 15) == 0;
}

}  // namespace

extern "C" int crn_bf3_operands(const float* packed, const int64_t* desc, int nlayers, int64_t total_blocks, void* out,
                                crnStream stream) {
  CRN_ENTRY(stream);
  if (!packed || !desc || !out || nlayers < 1 || total_blocks < 1 || total_blocks > 0x7fffffff) return CRN_EINVAL;
  hipLaunchKernelGGL(bf3_operands_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, packed,
                     reinterpret_cast<const long long*>(desc), nlayers, reinterpret_cast<char*>(out));
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

namespace {
long long* g_stamps = nullptr;
template <int T, int CB, int TW>
int launch_e2d(const E2dGeom& g, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_done = false;                      // > 64 KiB of dynamic LDS needs the attribute (once per kernel)
  if (!attr_done) {
    CRN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_e2d_kernel<T, CB, TW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_e2d_kernel<T, CB, TW>), grid, dim3(256), lds, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
}  // namespace

extern "C" int crn_conv2d_bf3(const crnView* x, const crnInTransform* tr, const void* wop, int Npad, const float* bias,
                              int bias_sB, const crnView* y, int kh, int kw, int ph, int pw, int accumulate,
                              crnStream stream) {
  if (!x || !y || !wop || x->B != y->B || x->B < 1) return CRN_EINVAL;
  if (!dense_2d(*x) || !dense_2d(*y)) return CRN_EINVAL;
  if ((x->C & 31) || (y->C & 63) || Npad != y->C) return CRN_EINVAL;
  if (x->D != y->D || x->H != y->H || x->W != y->W) return CRN_EINVAL;
  const bool k3 = kh == 3 && kw == 3, k1 = kh == 1 && kw == 1;
  if (!k3 && !k1) return CRN_EINVAL;
  if (k1 && (ph || pw)) return CRN_EINVAL;
  const int64_t S = (int64_t)x->D * x->H * x->W;
  if (S * x->C * 4 >= ((int64_t)1 << 31)) return CRN_EINVAL;           // one sample inside the 2 GiB buffer range
  hipStream_t st = (hipStream_t)stream;
  const bool armed = crn_splitk_take_armed();
  { const int rcf = crn_splitk_flush(st); if (rcf != CRN_OK) return rcf; }
  E2dGeom g{};
  g.x = *x; g.y = *y;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.wop = wop; g.bias = bias; g.bias_sB = bias_sB;
  g.ntn = Npad / 16; g.nblk = x->C / 32; g.ph = ph; g.pw = pw;
  static const int dbg = getenv("CRN_E2D_DBG") ? atoi(getenv("CRN_E2D_DBG")) : 0;
  g.dbg = dbg;
  if (dbg & 16) {
    if (!g_stamps) CRN_HIP(hipMalloc(&g_stamps, 32 * sizeof(long long)));
    g.stamps = g_stamps;
  }
  int TW = 16, CB = 1;
  int64_t tiles;
  if (k3) {
    if (x->D != 1 || ph < 0 || ph > 2 || pw < 0 || pw > 2) return CRN_EINVAL;
    TW = (x->W % 16 == 0) ? 16 : 8;
    const int TH = 64 / TW;
    if (x->W % TW || x->H % TH) return CRN_EINVAL;
    g.tilesW = x->W / TW; g.tilesH = x->H / TH;
    tiles = (int64_t)g.tilesW * g.tilesH;
    g.nchunks = g.nblk;
  } else {
    if (S % 64) return CRN_EINVAL;
    CB = 4;
    g.tilesW = (int)(S / 64); g.tilesH = 1;
    tiles = g.tilesW;
    g.nchunks = (g.nblk + CB - 1) / CB;
  }
  const int64_t base_blocks = tiles * x->B * (y->C / 64);
  static const int kFill = getenv("CRN_E2D_FILL") ? atoi(getenv("CRN_E2D_FILL")) : 256;
  static const int force_splits = getenv("CRN_E2D_SPLITS") ? atoi(getenv("CRN_E2D_SPLITS")) : 0;
  int splits = 1;
  while (base_blocks * splits < kFill && splits * 2 <= g.nchunks && splits < 16) splits *= 2;
  if (force_splits > 0) splits = std::min(force_splits, g.nchunks);
  g.chunks_per_split = (g.nchunks + splits - 1) / splits;
  splits = (g.nchunks + g.chunks_per_split - 1) / g.chunks_per_split;
  g.tab = g.tr.scale ? g.chunks_per_split * CB * 32 : 0;
  crnView yreal = *y;
  const float* scratch = nullptr;
  if (splits > 1) {
    float* sc = crn_splitk_scratch((size_t)splits * x->B * y->C * S, st);
    if (!sc) return CRN_ENOMEM;
    scratch = sc;
    g.y.base = sc; g.y.sC = S; g.y.sB = (int64_t)y->C * S; g.y.sH = x->W; g.y.sD = x->H * x->W;
    g.mode = 3;
  } else {
    g.mode = accumulate ? 1 : 0;
  }
  const size_t plane_bytes = (size_t)(k3 ? 4 * (128 + 4) : 16 * (64 + 4)) * 16;   // conv_e2d_kernel: PLANE
  const size_t lds = (size_t)g.tab * 8 + 4 * plane_bytes;
  if (lds > 160 * 1024) return CRN_EINVAL;
  const dim3 grid((unsigned)(tiles * x->B), (unsigned)(y->C / 64), (unsigned)splits);
  int rc;
  if (k1) rc = launch_e2d<1, 4, 16>(g, grid, lds, st);
  else if (TW == 16) rc = launch_e2d<9, 1, 16>(g, grid, lds, st);
  else rc = launch_e2d<9, 1, 8>(g, grid, lds, st);
  if (rc == CRN_OK && splits > 1) {
    if (armed && !accumulate && yreal.sB == (int64_t)yreal.C * S) {      // the next BatchRenorm launch adds them up
      crn_splitk_set_pending(yreal, scratch, splits, st);
      return rc;
    }
    rc = crn_splitk_reduce(yreal, scratch, splits, accumulate, st);
  }
  return rc;
}

// tuning aid (CRN_E2D_DBG=16): the shader-clock stamps of the last crn_conv2d_bf3 call (synchronises the device)
#ifdef CRN_TOOLS      // tools/_build/libcorenet_hip_tools.so only (corenet_amd.build.build_tools)
extern "C" int crn_e2d_debug_stamps(long long* out32) {
  if (!g_stamps) return CRN_EINVAL;
  CRN_HIP(hipDeviceSynchronize());
  CRN_HIP(hipMemcpy(out32, g_stamps, 32 * sizeof(long long), hipMemcpyDeviceToHost));
  return CRN_OK;
}
#endif

// The weight gradient of a 1x1 layer on the split-bf16 MFMA (see wgrad1x1_bf3_kernel); same contract as
// crn_conv_wgrad with a 1x1x1 window: dw[c*Npad + n] += sum T(x)[b,c,p] * dy[b,n,p], dw zeroed by the caller or by
// zero_first.  CRN_EINVAL for views that are not dense or positions per sample that are not a multiple of 32.
extern "C" int crn_conv_wgrad_1x1_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                                      int zero_first, crnStream stream) {
  CRN_ENTRY(stream);
  return crn_conv_wgrad_2d_bf3(x, tr, dy, dw, Npad, 1, 1, 0, 0, zero_first, stream);
}

extern "C" int crn_conv_wgrad_2d_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                                     int kh, int kw, int ph, int pw, int zero_first, crnStream stream) {
  CRN_ENTRY(stream);
  if (!x || !dy || !dw || Npad <= 0 || (Npad & 15) || x->B != dy->B || dy->C > Npad) return CRN_EINVAL;
  if (!dense_2d(*x) || !dense_2d(*dy)) return CRN_EINVAL;
  if (x->D != dy->D || x->H != dy->H || x->W != dy->W) return CRN_EINVAL;
  const int64_t S = (int64_t)x->D * x->H * x->W;
  if (S % 32 || x->sC != S || dy->sC != S) return CRN_EINVAL;
  if (S * x->C * 4 >= ((int64_t)1 << 31) || S * dy->C * 4 >= ((int64_t)1 << 31)) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (kh == 3 && kw == 3) {
    if (ph != 1 || pw != 1 || x->D != 1 || (x->W & 7)) return CRN_EINVAL;
    if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * 9 * Npad * 4, st));
    Wg3Geom g{};
    g.x = x->base; g.dy = dy->base; g.dw = dw;
    g.scale = tr ? tr->scale : nullptr; g.shift = tr ? tr->shift : nullptr;
    g.pre_relu = tr ? tr->pre_relu : 0; g.post_relu = tr ? tr->post_relu : 0;
    g.xsB = x->sB; g.dsB = dy->sB; g.C = x->C; g.N = dy->C; g.Npad = Npad; g.S = (int)S; g.H = x->H; g.W = x->W;
    g.ksteps = (int)((int64_t)x->B * S / 32);
    const int tiles = crn_cdiv(3 * x->C, 32) * crn_cdiv(dy->C, 32);
    static const int kFill3 = getenv("CRN_WG3_FILL") ? atoi(getenv("CRN_WG3_FILL")) : 512;
    int splits = std::max(1, std::min(g.ksteps / 8, crn_cdiv(kFill3, tiles)));
    if (crn_deterministic()) splits = 1;
    g.ksteps_per_block = crn_cdiv(g.ksteps, splits);
    splits = crn_cdiv(g.ksteps, g.ksteps_per_block);
    const dim3 grid((unsigned)crn_cdiv(3 * x->C, 32), (unsigned)crn_cdiv(dy->C, 32), (unsigned)splits);
    hipLaunchKernelGGL(wgrad3x3_bf3_kernel, grid, dim3(256), 0, st, g);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  if (kh != 1 || kw != 1 || ph || pw) return CRN_EINVAL;
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * Npad * 4, st));
  Wg1Geom g{};
  g.x = x->base; g.dy = dy->base; g.dw = dw;
  g.scale = tr ? tr->scale : nullptr; g.shift = tr ? tr->shift : nullptr;
  g.pre_relu = tr ? tr->pre_relu : 0; g.post_relu = tr ? tr->post_relu : 0;
  g.xsB = x->sB; g.dsB = dy->sB; g.C = x->C; g.N = dy->C; g.Npad = Npad; g.S = (int)S;
  g.ksteps = (int)((int64_t)x->B * S / 32);
  const int tiles = crn_cdiv(x->C, 64) * crn_cdiv(dy->C, 32);
  static const int kFill = getenv("CRN_WG1_FILL") ? atoi(getenv("CRN_WG1_FILL")) : 512;
  int splits = std::max(1, std::min(g.ksteps / 8, crn_cdiv(kFill, tiles)));       // >= 2 K steps per wave
  if (crn_deterministic()) splits = 1;
  g.ksteps_per_block = crn_cdiv(g.ksteps, splits);
  splits = crn_cdiv(g.ksteps, g.ksteps_per_block);
  const dim3 grid((unsigned)crn_cdiv(x->C, 64), (unsigned)crn_cdiv(dy->C, 32), (unsigned)splits);
  hipLaunchKernelGGL(wgrad1x1_bf3_kernel, grid, dim3(256), 0, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
