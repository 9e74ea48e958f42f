// bz_dev.h -- device-side helpers shared by the kernel files (bz_kernels.hip, bz_persist.hip): number formats, the fixed-point accumulator grid,
// the signed-nibble activation planes and the V_DOT8_I32_I4 group arithmetic of the int4 path, K/V row addressing.  Not part of the ABI.
#pragma once
#include "bz_internal.h"
#include <math.h>

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// streamed-once weights: non-temporal loads (guide: nt on weights that one CU reads once)
__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ldnt(const float4* p) {
  const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_round(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x;  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float round_act(float x, int act) {
  if (act == BZ_F16) return f16_round(x);
  if (act == BZ_BF16) return bf16_round(x);
  return x;
}
// Fixed point of the split-K accumulators: 2^-44 units when the activations are f16, 2^-32 otherwise.  A workgroup rounds its partial sum to the grid once
// (d2fix / f2fix), so a sum over a few hundred workgroups carries ~1e-12 of absolute error at 2^-44 against ~4e-9 at 2^-32 -- the latter is 1e-5 of an f16
// rounding interval of a value near 0.1, i.e. about one flipped f16 rounding per ten thousand outputs, and one flip in a layer's input is ~2000 one-ulp
// differences in its output.  Range at 2^-44: +-2^19 = 524288, enough where every value that is ever stored ends at 65504 (f16); bf16 / f32 activations have
// no such bound (the synthetic Mistral Q4_K_M weights drive intermediate values past 5e5), so they keep 2^-32 (+-2^31).  Producer and consumer of an
// accumulator belong to the same model and pass the same activation dtype.
// bf16 activations (round 3, with the exact dense sums): 2^-40, range +-2^23 -- a bf16 rounding interval is 2^3 wider than an f16 one, and at 2^-32 a sum over a few hundred
// partials still flipped ~0.1 bf16 roundings per token of a 16-layer model (each one puts that token's logits back at the bf16 noise floor).
__device__ __forceinline__ double fix_scale(int act) { return act == BZ_F16 ? 17592186044416.0 : (act == BZ_BF16 ? 1099511627776.0 : 4294967296.0); }            // 2^44 : 2^40 : 2^32
__device__ __forceinline__ double fix_inv(int act) { return act == BZ_F16 ? 5.6843418860808015e-14 : (act == BZ_BF16 ? 9.094947017729282e-13 : 2.3283064365386963e-10); }
__device__ __forceinline__ float fix2f(long long a, int act) {
  // ONE rounding of the exact fixed-point sum to f32 (the oracle rounds its double sum to f32 once): both 32-bit halves are exact in
  // double, so is their join below 2^53 (above, the join itself rounds to 53 bits first), and the cast rounds to nearest even
  const unsigned long long m = a < 0 ? (unsigned long long)(-a) : (unsigned long long)a;
  const double d = fma((double)(unsigned)(m >> 32), 4294967296.0, (double)(unsigned)(m & 0xffffffffull)) * fix_inv(act);
  const float r = (float)d;
  return a < 0 ? -r : r;
}
__device__ __forceinline__ long long f2fix(float p, int act) { return __float2ll_rn(p * (float)fix_scale(act)); }
__device__ __forceinline__ float vsrc_get(const VSrc& s, int i, int act) {
  if (s.fix) return round_act(fix2f(((const long long*)s.p)[i], act), act);
  return ((const float*)s.p)[i];
}
__device__ __forceinline__ float silu_f(float x) { return div_rn(x, 1.0f + bz_expf(-x)); }


// ---------------------------------------------------------------------------------------------------------
// activation slice -> signed-nibble planes + per-group parameters   (QG = 128 k per group, 16 lanes x 8 each)
//   x = c * xi exactly, c = 2^-e with am * 2^e in [2^29, 2^30) (am = the group's maximum magnitude): a power-of-two scale, so xi is the
//   activation itself.  xi is a 32-bit code = EIGHT balanced nibbles n_p in [-8, 7], xi = sum_p 16^p n_p.  f16 values carry 11 significant bits, so an
//   element within 2^-11 of the group maximum has nothing below bit 8: the two LOW planes (nibbles 0, 1) are zero for almost every group and are only
//   multiplied when a group's flag says they are not (wave-uniform branch; ~10 % of the groups of a normalised row).  Round 2 stopped at 24 bits (six
//   planes): elements below 2^-11.5 of their group maximum lost their last bits -- 3 elements of a 4096 row, an error of ~5e-9 of the output, i.e. one
//   flipped f16 rounding per ~70 k outputs: about one per layer, and one flip in a layer's input is ~2000 one-ulp differences in its output
//   (profiles/r03_parity_depth_*.txt).  With eight planes every f16 element down to 2^-19.5 of its group maximum is exact.
//   The weights are signed nibbles (q - 8) as well, so a 32-bit weight word meets a 32-bit plane word in ONE V_DOT8_I32_I4 -- six per eight weights
//   (eight in a flagged group), no unpacking at all (round 1: three int8 planes on V_DOT4_I32_I8 = six dots + two ANDs per eight weights).
//   Plane order in LDS and in registers: index 0..5 = the MAIN planes (nibbles 2..7 of the code), index 6, 7 = the LOW planes (nibbles 0, 1).
//   LDS image: pl[(g*4 + c)*8 + p] = uint4, the four plane-p words for the four weight words of 32-k chunk c of group g; nibble i of a word
//   is k offset (i >> 1) + 4 (i & 1), the order of the weight words (repack kernels below).
//   gpar[2g] = { c (float bits), S_0, S_1, S_2 }   gpar[2g+1] = { S_3, S_4, S_5, low }     S_p = sum of plane p over the group (main planes),
//   low = S_6 (12 bits) | S_7 (12 bits) << 12 | flag << 24   (flag: some low-plane nibble of the group is not zero)
//   sum_k (q_k - z) x_k = c * [ 256 * sum_{p<6} 16^p (D_p + (8 - z) S_p) + (D_6 + (8 - z) S_6) + 16 (D_7 + (8 - z) S_7) ],  D_p = sum_k (q_k - 8) n_p,k
//   -- all of it exact integer arithmetic.
// ---------------------------------------------------------------------------------------------------------
#define XQ_NP 8   // planes stored per chunk
#define XQ_NM 6   // main planes (always multiplied)
struct OpOr { __device__ __forceinline__ static int f(int a, int b) { return a | b; } };
__device__ __forceinline__ int xq_pack_low(int s6, int s7, int flag) { return (s6 & 0xFFF) | ((s7 & 0xFFF) << 12) | (flag ? (1 << 24) : 0); }
// wave-uniform: does group `gi` (index of its gpar pair) have non-zero low planes?  (every lane reads the same LDS word)
__device__ __forceinline__ bool xq_low(const int4* gpar, int gi) { return __builtin_amdgcn_readfirstlane(gpar[gi + 1].w >> 24) != 0; }

// one thread: 8 consecutive activations -> one word per plane (+ the plane sums); am = group maximum (already reduced).
// The balanced digits of xi are the unsigned base-16 digits of xi + 0x88888888 minus 8, i.e. in two's complement simply
// code = (xi + 0x88888888) ^ 0x88888888: nibble p of the 32-bit code IS the stored nibble of plane p.  What remains is an 8 x 8 nibble
// transpose: pairs (k, k + 4) are interleaved into bytes with three mask ops, the bytes gathered per plane with V_PERM_B32.
__device__ __forceinline__ void xq_split8(const float (&v)[8], float am, unsigned (&w)[XQ_NP], int (&sp)[XQ_NP], float& cscale) {
  const unsigned eb = (__float_as_uint(am) >> 23) & 255u;               // biased exponent of the group maximum
  const bool live = eb >= 32u && eb < 255u;
  const float inv = live ? __uint_as_float((283u - eb) << 23) : 0.f;    // 2^(29 - (eb - 127))
  cscale = live ? __uint_as_float((eb - 29u) << 23) : 0.f;              // its reciprocal
  unsigned code[8];
#pragma unroll
  for (int i = 0; i < 8; i++) code[i] = ((unsigned)(int)rintf(v[i] * inv) + 0x88888888u) ^ 0x88888888u;
  unsigned E[4], O[4];          // byte k of E[i] / O[i]: nibble 2k / 2k + 1 of the codes, k offsets i (low nibble of the byte) and i + 4 (high nibble)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    E[i] = ((code[i + 4] & 0x0F0F0F0Fu) << 4) | (code[i] & 0x0F0F0F0Fu);
    O[i] = ((code[i] >> 4) & 0x0F0F0F0Fu) | (code[i + 4] & 0xF0F0F0F0u);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned sel = 0x0c0c0000u | ((4u + k) << 8) | (unsigned)k;   // [lo.byte k, hi.byte k, 0, 0]
    const int pe = k == 0 ? 6 : 2 * k - 2, po = k == 0 ? 7 : 2 * k - 1; // nibbles 0, 1 are the low planes (index 6, 7); nibble n >= 2 is main plane n - 2
    w[pe] = __builtin_amdgcn_perm(E[1], E[0], sel) | (__builtin_amdgcn_perm(E[3], E[2], sel) << 16);
    w[po] = __builtin_amdgcn_perm(O[1], O[0], sel) | (__builtin_amdgcn_perm(O[3], O[2], sel) << 16);
  }
#pragma unroll
  for (int p = 0; p < XQ_NP; p++) sp[p] = __builtin_amdgcn_sdot8((int)w[p], 0x11111111, 0, false);   // sum of the eight signed nibbles
}
// the second parameter word of a group from one lane's reduced sums (the flag: OR over the group's lanes of "my low words are not zero")
template <int NL>
__device__ __forceinline__ int4 xq_gpar_hi(const unsigned (&w)[XQ_NP], const int (&sp)[XQ_NP]) {
  const int fl = grp_reduce<NL, OpOr>((int)((w[6] | w[7]) != 0u));
  return make_int4(sp[3], sp[4], sp[5], xq_pack_low(sp[6], sp[7], fl));
}

template <int NTH>
__device__ __forceinline__ void quant_x128(const float* xs, int KR, uint4* pl, int4* gpar) {
  unsigned* plw = (unsigned*)pl;
  for (int base = 0; base < KR; base += NTH * 8) {
    const int e0 = base + threadIdx.x * 8;
    const bool on = e0 < KR;
    float v[8];
    if (on) {
      float4 a = *(const float4*)(xs + e0), b = *(const float4*)(xs + e0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = 0.f;
    }
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) am = fmaxf(am, fabsf(v[i]));
    am = grp_reduce<16, OpMax>(am);
    unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
    xq_split8(v, am, w, sp, cs);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<16, OpAdd>(sp[p]);
    const int4 g2w = xq_gpar_hi<16>(w, sp);
    if (on) {
      const int word = ((e0 >> 5) * XQ_NP) * 4 + ((e0 >> 3) & 3);       // chunk e0 / 32, weight word (e0 / 8) % 4
#pragma unroll
      for (int p = 0; p < XQ_NP; p++) plw[word + p * 4] = w[p];
      if ((threadIdx.x & 15) == 0) {
        gpar[2 * (e0 >> 7)] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
        gpar[2 * (e0 >> 7) + 1] = g2w;
      }
    }
  }
}

// 64 values -> planes + parameters (one 64-k half group = two chunks; threads 0..7 of the block, 8 values each)
__device__ __forceinline__ void quant_x64(const float* a, uint4* pl, int4* gpar) {
  const int t = threadIdx.x;
  if (t >= 64) return;                       // first wave only (the reductions below stay inside it)
  const bool on = t < 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = on ? a[t * 8 + i] : 0.f;
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) am = fmaxf(am, fabsf(v[i]));
  am = grp_reduce<8, OpMax>(am);
  unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
  xq_split8(v, am, w, sp, cs);
#pragma unroll
  for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<8, OpAdd>(sp[p]);
  const int4 g2w = xq_gpar_hi<8>(w, sp);
  if (on) {
    unsigned* plw = (unsigned*)pl;
    const int word = ((t >> 2) * XQ_NP) * 4 + (t & 3);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) plw[word + p * 4] = w[p];
    if (t == 0) {
      gpar[0] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
      gpar[1] = g2w;
    }
  }
}


// one 32-k chunk: the lane's 16-byte weight piece against the six main planes (24 V_DOT8_I32_I4, 6 broadcast ds_read_b128)
__device__ __forceinline__ void q4_chunk(const uint4& w, const uint4* pl8, int (&D)[XQ_NM]) {
  const unsigned W[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int p = 0; p < XQ_NM; p++) {
    const uint4 P = pl8[p];
    const unsigned X[4] = {P.x, P.y, P.z, P.w};
#pragma unroll
    for (int j = 0; j < 4; j++) D[p] = __builtin_amdgcn_sdot8((int)W[j], (int)X[j], D[p], false);
  }
}
// ... against the two LOW planes: a second, rare pass over a flagged group's chunks (its weights are still in registers)
__device__ __forceinline__ void q4_chunk_low(const uint4& w, const uint4* pl8, int (&DL)[2]) {
  const unsigned W[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const uint4 P = pl8[XQ_NM + p];
    const unsigned X[4] = {P.x, P.y, P.z, P.w};
#pragma unroll
    for (int j = 0; j < 4; j++) DL[p] = __builtin_amdgcn_sdot8((int)W[j], (int)X[j], DL[p], false);
  }
}
// the same main planes against two weight pieces (gate and up of the fused MLP): the planes are read from LDS once
__device__ __forceinline__ void q4_chunk2(const uint4& wa, const uint4& wb, const uint4* pl8, int (&Da)[XQ_NM], int (&Db)[XQ_NM]) {
  const unsigned Wa[4] = {wa.x, wa.y, wa.z, wa.w}, Wb[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
  for (int p = 0; p < XQ_NM; p++) {
    const uint4 P = pl8[p];
    const unsigned X[4] = {P.x, P.y, P.z, P.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      Da[p] = __builtin_amdgcn_sdot8((int)Wa[j], (int)X[j], Da[p], false);
      Db[p] = __builtin_amdgcn_sdot8((int)Wb[j], (int)X[j], Db[p], false);
    }
  }
}
// low part of a flagged group for one column: (D_6 + (8 - z) S_6) + 16 (D_7 + (8 - z) S_7)
__device__ __forceinline__ int q4_low(const int (&DL)[2], const int4 g2, int z) {
  const int zz = 8 - z, s6 = (g2.w << 20) >> 20, s7 = (g2.w << 8) >> 20;
  return (DL[0] + zz * s6) + ((DL[1] + zz * s7) << 4);
}
// group epilogue: s * c * [256 * sum_{p<6} 16^p (D_p + (8 - z) S_p) + lw], exact -- |V_p| <= 2^14, the two three-plane halves fit int32, their join
// (< 2^36), its shift by 8 bits plus the low part lw (< 2^19; 0 for an unflagged group) and the product with the f32 factor s c are exact in double
__device__ __forceinline__ double q4_term(const int (&D)[XQ_NM], int lw, const int4 g1, const int4 g2, float s, int z) {
  const int zz = 8 - z;
  const int V0 = D[0] + zz * g1.y, V1 = D[1] + zz * g1.z, V2 = D[2] + zz * g1.w, V3 = D[3] + zz * g2.x, V4 = D[4] + zz * g2.y, V5 = D[5] + zz * g2.z;
  const int lo = V0 + (V1 << 4) + (V2 << 8), hi = V3 + (V4 << 4) + (V5 << 8);
  return (double)(s * __int_as_float(g1.x)) * fma(fma((double)hi, 4096.0, (double)lo), 256.0, (double)lw);
}

// NCH 32-k chunks w[OFF .. OFF + NCH) of one tile (a register array: indexed with compile-time constants only -- a pointer into it would send the array
// to scratch memory) whose planes start at chunk index co; group parameters at gpar[gp], gpar[gp + 1].
// A flagged group takes a SEPARATE straight-line path (main + low planes chunk by chunk): a conditional second pass over the weights after the main pass
// keeps weights and accumulators alive across a branch, and the register allocator of the fused kernels answered that with 264 spilled registers.
// SB: a scheduling barrier after every chunk -- without it the scheduler of a large kernel hoists the plane reads of ALL chunks (24 registers each) to the top
template <int OFF, int NCH, int NTOT, bool SB = false>
__device__ __forceinline__ void q4g_consume_at(const uint4 (&w)[NTOT], int co, int gp, const uint4* pl, const int4* gpar, float s, int z, double& y) {
  int D[XQ_NM] = {0, 0, 0, 0, 0, 0};
  if (!xq_low(gpar, gp)) {
#pragma unroll
    for (int c = 0; c < NCH; c++) { q4_chunk(w[OFF + c], pl + (co + c) * XQ_NP, D); if (SB) __builtin_amdgcn_sched_barrier(0); }
    y += q4_term(D, 0, gpar[gp], gpar[gp + 1], s, z);
  } else {
    int DL[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < NCH; c++) { q4_chunk(w[OFF + c], pl + (co + c) * XQ_NP, D); q4_chunk_low(w[OFF + c], pl + (co + c) * XQ_NP, DL); if (SB) __builtin_amdgcn_sched_barrier(0); }
    y += q4_term(D, q4_low(DL, gpar[gp + 1], z), gpar[gp], gpar[gp + 1], s, z);
  }
}
template <int NCH>
__device__ __forceinline__ void q4g_consume_n(const uint4 (&w)[NCH], int co, int gp, const uint4* pl, const int4* gpar, float s, int z, double& y) {
  q4g_consume_at<0, NCH, NCH>(w, co, gp, pl, gpar, s, z, y);
}
// one 128-k group (four chunks) of one tile
__device__ __forceinline__ void q4g_consume(const uint4 (&w)[4], int g, const uint4* pl, const int4* gpar, float s, int z, double& y) {
  q4g_consume_at<0, 4, 4>(w, g * 4, 2 * g, pl, gpar, s, z, y);
}
// two weight tiles against the SAME activation group (gate and up of the fused MLP): chunks wa[OA .. OA + 4), wb[OB .. OB + 4) (wa and wb may be one array)
template <int OA, int OB, bool SB, int NA, int NB>
__device__ __forceinline__ void q4g_consume2_x(const uint4 (&wa)[NA], const uint4 (&wb)[NB], int g, const uint4* pl, const int4* gpar, float sa, int za, float sb, int zb,
                                               double& ya, double& yb) {
  int Da[XQ_NM] = {0, 0, 0, 0, 0, 0}, Db[XQ_NM] = {0, 0, 0, 0, 0, 0};
  const int4 g1 = gpar[2 * g], g2 = gpar[2 * g + 1];
  if (!xq_low(gpar, 2 * g)) {
#pragma unroll
    for (int c = 0; c < 4; c++) { q4_chunk2(wa[OA + c], wb[OB + c], pl + (g * 4 + c) * XQ_NP, Da, Db); if (SB) __builtin_amdgcn_sched_barrier(0); }
    ya += q4_term(Da, 0, g1, g2, sa, za);
    yb += q4_term(Db, 0, g1, g2, sb, zb);
  } else {
    int La[2] = {0, 0}, Lb[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      q4_chunk2(wa[OA + c], wb[OB + c], pl + (g * 4 + c) * XQ_NP, Da, Db);
      q4_chunk_low(wa[OA + c], pl + (g * 4 + c) * XQ_NP, La); q4_chunk_low(wb[OB + c], pl + (g * 4 + c) * XQ_NP, Lb);
      if (SB) __builtin_amdgcn_sched_barrier(0);
    }
    ya += q4_term(Da, q4_low(La, g2, za), g1, g2, sa, za);
    yb += q4_term(Db, q4_low(Lb, g2, zb), g1, g2, sb, zb);
  }
}
template <int OFF, int NTOT>
__device__ __forceinline__ void q4g_consume2_at(const uint4 (&wa)[NTOT], const uint4 (&wb)[NTOT], int g, const uint4* pl, const int4* gpar, float sa, int za, float sb, int zb,
                                                double& ya, double& yb) {
  q4g_consume2_x<OFF, OFF, false>(wa, wb, g, pl, gpar, sa, za, sb, zb, ya, yb);
}
template <int OA, int OB, int NTOT, bool SB = false>
__device__ __forceinline__ void q4g_consume2_ab(const uint4 (&w)[NTOT], int g, const uint4* pl, const int4* gpar, float sa, int za, float sb, int zb, double& ya, double& yb) {
  q4g_consume2_x<OA, OB, SB>(w, w, g, pl, gpar, sa, za, sb, zb, ya, yb);
}

// fixed point from the double a lane accumulated over its groups
__device__ __forceinline__ long long d2fix(double p, int act) { return __double2ll_rn(p * fix_scale(act)); }

template <int ACT> __device__ __forceinline__ float round_t(float x) {
  if (ACT == BZ_F16) return f16_round(x);
  if (ACT == BZ_BF16) return bf16_round(x);
  return x;
}
#define DPP_ROR4 0x124   // row_ror:4 / row_ror:8: rotate within a 16-lane row (sums that must not mix even and odd lanes)
#define DPP_ROR8 0x128


__device__ __forceinline__ void lds_wait_count(volatile unsigned* cnt, unsigned n) {
  while (*cnt < n) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");   // the LDS reads that follow stay behind the wait
}

__device__ __forceinline__ size_t kv_row_off(const KvView& kv, int layer, int kvh, int p) {
  if (kv.paged) {
    const int blk = kv.block_table[p / kv.bs];
    return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + (p % kv.bs)) * kv.hd;
  }
  return (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd;
}
__device__ __forceinline__ size_t kv_slot_off(const KvView& kv, int layer, int kvh, int slot) {
  const int blk = slot / kv.bs, o = slot % kv.bs;
  return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + o) * kv.hd;
}

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
template <int KVDT>
__device__ __forceinline__ float dot2acc(unsigned a, unsigned b, float c) {
  if (KVDT == BZ_F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, a), __builtin_bit_cast(b2_t, b), c, false);
}
template <int KVDT>
__device__ __forceinline__ unsigned pack2(float x0, float x1) {
  if (KVDT == BZ_F16) return (unsigned)__half_as_ushort(f16_cvt(x0)) | ((unsigned)__half_as_ushort(f16_cvt(x1)) << 16);
  return (__float_as_uint(bf16_round(x0)) >> 16) | (__float_as_uint(bf16_round(x1)) & 0xffff0000u);
}
template <int KVDT>
__device__ __forceinline__ void unpack2(unsigned u, float& x0, float& x1) {
  if (KVDT == BZ_F16) { x0 = __half2float(__ushort_as_half((unsigned short)(u & 0xffffu))); x1 = __half2float(__ushort_as_half((unsigned short)(u >> 16))); }
  else { x0 = __uint_as_float(u << 16); x1 = __uint_as_float(u & 0xffff0000u); }
}

// branch-free accessors for the activation source (f32 or 2^-32 fixed point): both loads are unconditional so that the compiler can batch them
__device__ __forceinline__ void vsrc_issue(const void* p, int fix, int i, unsigned& lo, unsigned& hi) {
  const unsigned* u = (const unsigned*)p;
  const size_t e = (size_t)i << fix;
  lo = u[e]; hi = u[e + fix];
}
__device__ __forceinline__ float vsrc_finish(int fix, unsigned lo, unsigned hi, int act) {
  const float f = round_act(fix2f((long long)(((unsigned long long)hi << 32) | lo), act), act);
  return fix ? f : __uint_as_float(lo);
}
template <int PAGED>
__device__ __forceinline__ size_t kv_row_off_t(const KvView& kv, int layer, int kvh, int p) {
  if (PAGED) {
    const int blk = kv.block_table[p / kv.bs];
    return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + (p % kv.bs)) * kv.hd;
  }
  return (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd;
}

