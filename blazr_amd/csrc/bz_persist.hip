// bz_persist.hip -- the Llama decode layers of one step as ONE persistent launch (int4 AWQ / GPTQ-without-act-order weights, f16 activations,
// hidden 4096, 32 query / 8 kv heads x 128: the Llama-3-8B shape of BASELINE.json configs[1]).
//
// Why: the step used to be three dependent launches per layer (q/k/v GEMV -> attention + o_proj -> fused MLP).  Each launch pays launch -> first
// load, a first-load latency before any dot product can start, the drain of its atomics and the kernel boundary: ~4-6 us of fixed cost against
// 2-15 us of streaming.  The weights of the NEXT phase do not depend on anything the current phase computes, so here every wave requests its
// next-phase weights BEFORE it waits at the phase boundary: HBM keeps streaming through the boundary, and what is left on the dependent chain is the
// hand-off itself.  One workgroup of 16 waves per CU (256 workgroups), three phases per layer separated by a grid barrier:
//   Q  h' = R(h + R(mlp_prev)), RMSNorm, q/k/v GEMV       unit = (64-column tile, 256-k slice), 6 tile waves per workgroup        -> ring_q (fixed point)
//   A  q/k/v finish + RoPE + KV append + attention + o_proj workgroup = (head, 1/8 of the output columns), as k_attn2<FUSE>         -> ring_o
//   M  h'' = R(h' + R(o)), RMSNorm, gate/up, SiLU*up, down  workgroup = 64 intermediate columns (I/64 of the 256 workgroups), as k_mlp_q4g -> ring_m
// The arithmetic is the launch-per-phase kernels' (bz_dev.h: the same planes, group terms and fixed-point grid; integer atomics are associative), so
// the persistent step is BIT-IDENTICAL to the three-launch step -- which is how it is tested (tests/test_gpu_persist.py).
//
// Data that crosses workgroups: only the three fixed-point accumulators.  Producers add with agent-scope 64-bit atomics (performed at the memory side),
// every wave waits for its atomics' acknowledgements (s_waitcnt vmcnt(0)) before its workgroup arrives at the barrier; consumers read them with agent-scope
// atomic loads (global_load_dwordx2 sc1: never served from a stale L1 / non-coherent L2 line) after the barrier -- the "8-byte agent atomics on both sides" form
// of MI355X_MICROARCH.md (inter-workgroup visibility).  An accumulator is zeroed (atomic AND 0) in the phase after the one that read it.  The residual
// stream h never leaves the CU: every workgroup keeps its own copy in LDS and applies the same updates to it (reading 32 KB of accumulator per phase from L2
// is what every workgroup of the launch-per-phase kernels did as well).
//
// Barrier: one control wave per workgroup (wave 15) arrives on a per-group counter (blockIdx & 7: 32 arrivals), the last arrival of a group on a top
// counter, the last group bumps a generation word that all control waves poll (sc1 loads, s_sleep).  Counters are monotonic (no reset, wrap-safe).  Every
// spin has a wall-clock limit (s_memrealtime): a workgroup that waits longer than 20 ms sets the error word and stops waiting, so a lost arrival ends the
// launch with an error code instead of hanging the device.
//
// Reference anchor for what one step must compute: /root/reference/src/engine/cuda_graphs.rs:97-170 (forward_graph_mode), executor_generate.rs:357,372.
#include "bz_internal.h"
#include "bz_dev.h"
#include <string.h>

namespace {

constexpr int PH = 4096, PG = PH / 128, PNQ = 32, PNKV = 8, PHD = 128, PNT_QKV = (PNQ + 2 * PNKV) * PHD / 64;   // 96 q/k/v tiles
constexpr int BAR_STRIDE = 32;   // unsigned words between barrier counters (128 B: one line each)
constexpr int BAR_TOP = 8 * BAR_STRIDE, BAR_GEN = 9 * BAR_STRIDE, BAR_ERR = 10 * BAR_STRIDE;

__device__ __forceinline__ unsigned long long ld_acc(const long long* p) {
  return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 sc1
}
__device__ __forceinline__ unsigned ld_word(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void acc_add(long long* p, long long v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
__device__ __forceinline__ void acc_zero(long long* p) { (void)__hip_atomic_fetch_and((unsigned long long*)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long now100() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_s_waitcnt(0xc07f); }                  // lgkmcnt(0): this wave's LDS stores are done
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // every global op of this wave (atomics included) is acknowledged

// LDS counters are monotonic; a waiter knows the value it waits for.  The limit keeps a lost signal from hanging the CU.
__device__ __forceinline__ void lds_signal(volatile unsigned* c) { lds_fence(); atomicAdd((unsigned*)c, 1u); }
__device__ __forceinline__ bool lds_wait(volatile unsigned* c, unsigned target) {
  if (*c >= target) { asm volatile("" ::: "memory"); return true; }
  const unsigned long long t0 = now100();
  while (*c < target) {
    __builtin_amdgcn_s_sleep(1);
    if (now100() - t0 > 2000000ull) return false;   // 20 ms
  }
  asm volatile("" ::: "memory");
  return true;
}

struct PersistLds {
  float hres[PH];                 // the residual stream (values representable in f16)
  uint4 xpl[(PH / 32) * XQ_NP];   // planes of the normalised row (M: the whole row; Q: the first 8 chunks hold the workgroup's 256-k slice)
  int4 gpar[2 * PG];
  double dred[16];
  unsigned cnt[16];               // monotonic LDS counters (indices below)
  union {
    struct { double part[8 * 128]; uint4 apl[2 * XQ_NP]; int4 apar[2]; } m;
    struct { unsigned q2[64], k2[64], v2[64]; double pout[8 * 128]; double lred[8]; float wred[8]; float outh[128]; uint4 xpl[4 * XQ_NP]; int4 gpar[2]; } at;
  } u;
};
enum { C_SS = 0, C_PL = 1, C_ARR = 2, C_REL = 3, C_QKV = 4, C_MAX = 5, C_PV = 6, C_OUT = 7, C_QA = 8, C_PART = 9, C_TAIL = 10 };

}  // namespace

struct PLayer {
  const uint4* Wqkv; const __half* Sqkv; const unsigned char* Zqkv;
  const uint4* Wo; const __half* So; const unsigned char* Zo;
  const uint4* Wgu; const __half* Sgu; const unsigned char* Zgu;
  const uint4* Wd; const __half* Sd; const unsigned char* Zd;
  const float* attn_norm; const float* ffn_norm;
};
struct PersistArgs {
  const PLayer* layers; int n_layers;
  const float* h_in;      // [H] residual stream entering layer 0 of the table (the embedding row, or the pieces API's hidden row with prev already added)
  float* h_out;           // [H] residual stream the head kernel continues from: h'' of the last layer (its MLP output stays in ring_m)
  long long* ring_m; long long* ring_q; long long* ring_o;   // fixed-point accumulators (zero on entry): MLP out [H], q/k/v [6144], o_proj out [H]
  const float* rope_cur; const int* pos; KvView kv;
  unsigned* bar;          // barrier words (persist across launches): 8 group counters, top counter, generation, error -- BAR_STRIDE words apart
  unsigned* err_host;     // host-pinned word: set (never cleared by the device) when a barrier wait ran into its limit
  float eps; int I;
  long long* stamps;      // diagnostic (BZ_PERSIST_STAMPS): s_memrealtime of workgroup 0 / 131 at the phase boundaries of layer 1 (nullptr: off)
};

// ---------------------------------------------------------------------------------------------------------
// 512 threads = 8 waves per workgroup (2 per SIMD: 256 VGPRs per thread -- the prefetch registers of a phase live through the previous phase's tail,
// the barrier and the next prologue, and with 16 waves' 128 registers the allocator spilled 600 of them).
//   row update (Q and M): thread t owns the octet t of the row (512 x 8 = 4096): h <- R(h + R(acc)), sum of squares over the 8 waves, then the planes
//   Q: waves 0..5 own one 64-column tile x the workgroup's 256-k slice each (8 chunks = 8 KiB)
//   A: wave w owns positions 32 w .. 32 w + 31 of every 256-position chunk (8 wave-wide loads of 4 rows), and one o_proj tile of the head's slab
//   M: wave w owns k-groups 4 w .. 4 w + 3 of gate and up (32 chunks) and 8 output tiles of the down slab; control wave = 7
// ---------------------------------------------------------------------------------------------------------
template <int PAGED>
__global__ __launch_bounds__(512) void k_llama_persist(PersistArgs a_in) {
  const PersistArgs& a = a_in;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PersistLds& S = *(PersistLds*)smem_raw;
  volatile unsigned* cnt = S.cnt;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wg = blockIdx.x;
  constexpr int ACT = BZ_F16;
  constexpr int NW = 8, CW = NW - 1;      // waves; control wave
  const int I = a.I, NTI = I >> 6, NUNIT = I >> 6;
  const KvView& kv = a.kv;

  // ---- start: LDS counters, the residual stream, the position, the barrier generation ------------------------------------
  if (tid < 16) S.cnt[tid] = 0;
  for (int i = tid; i < PH / 4; i += NW * 64) ((float4*)S.hres)[i] = ((const float4*)a.h_in)[i];
  const int pos = __builtin_amdgcn_readfirstlane(a.pos[0]);
  const int len = pos + 1, pmax = pos > 0 ? pos - 1 : 0;
  unsigned gen = 0;
  if (wave == CW) gen = __builtin_amdgcn_readfirstlane(ld_word(a.bar + BAR_GEN));
  __syncthreads();
  unsigned e_ss = 0, e_pl = 0, e_arr = 0, e_rel = 0, e_qkv = 0, e_max = 0, e_pv = 0, e_out = 0, e_qa = 0, e_part = 0, e_tail = 0;   // expected counter values (wave-uniform)
  bool dead = false;   // a wait ran into its limit: skip every later wait (the launch ends with the error word set)

  // registers that carry prefetched weights across phase boundaries: 64 per thread (16 KiB per wave, 128 KiB per workgroup: ~5 us of the chip's stream).
  // Q: P[0..7] the tile wave's 8 chunks.  A: P[0..3] the o_proj slab, P[8..15] the K rows of the first chunk.
  // M: P[0..7] gate, P[8..15] up of the wave's FIRST TWO groups.
  uint4 P[16];
  float sA[4], sB[4]; int zA[4], zB[4];
  float4 nwa = make_float4(0, 0, 0, 0), nwb = make_float4(0, 0, 0, 0);   // norm weights of this thread's octet in the next row update

  // ---- role constants ----------------------------------------------------------------------------------------------------
  const int oct = tid, i0 = oct * 8;                               // row update: this thread's octet
  const int q_slice = wg & 15, q_tg = wg >> 4;                     // Q: 256-k slice and tile group of this workgroup; tile waves 0..5
  const bool q_tilewave = wave < 6;
  const int q_tile = q_tg * 6 + (wave < 6 ? wave : 0);             // < 96 by construction (16 groups x 6)
  const int q_osl = 32 * q_slice;                                  // the slice's 32 octets: threads q_osl .. q_osl + 31 (one half of wave q_osl / 64)
  const int a_hq = wg >> 3, a_cs = wg & 7, a_kvh = a_hq >> 2;      // A: head and column slice
  const int piece = lane & 15, rsub = lane >> 4;
  const bool m_on = wg < NUNIT;                                    // M: unit (64 intermediate columns)
  const int m_sl = m_on ? wg : 0;

#define PSTAMP(i) do { if (a.stamps && l == 1 && lane == 0 && (wg == 0 || wg == 131)) a.stamps[((wg ? 1 : 0) * 8 + wave) * 32 + (i)] = (long long)now100(); } while (0)
#define GRID_BARRIER()                                                                                                               \
  do {                                                                                                                               \
    vm_drain();                                                                                                                      \
    if (lane == 0) lds_signal(&cnt[C_ARR]);                                                                                          \
    e_arr += NW; e_rel += 1;                                                                                                         \
    if (wave == CW) {                                                                                                                \
      if (!dead && !lds_wait(&cnt[C_ARR], e_arr)) dead = true;                                                                       \
      if (lane == 0 && !dead) {                                                                                                      \
        const unsigned o1 = atomicAdd(a.bar + (wg & 7) * BAR_STRIDE, 1u);                                                            \
        if ((o1 & 31u) == 31u) {                                                                                                     \
          const unsigned o2 = atomicAdd(a.bar + BAR_TOP, 1u);                                                                        \
          if ((o2 & 7u) == 7u) atomicAdd(a.bar + BAR_GEN, 1u);                                                                       \
        }                                                                                                                            \
        const unsigned long long t0 = now100();                                                                                     \
        bool ok = true;                                                                                                              \
        while (ld_word(a.bar + BAR_GEN) == gen) {                                                                                    \
          __builtin_amdgcn_s_sleep(2);                                                                                               \
          if (now100() - t0 > 2000000ull || ld_word(a.bar + BAR_ERR) != 0u) { ok = false; break; }                                   \
        }                                                                                                                            \
        if (!ok) { atomicExch(a.bar + BAR_ERR, 1u); *(volatile unsigned*)a.err_host = 1u; }                                          \
      }                                                                                                                              \
      gen += 1;                                                                                                                      \
      if (lane == 0) { S.cnt[C_REL] = e_rel; }                                                                                       \
      lds_fence();                                                                                                                   \
    } else {                                                                                                                         \
      if (!dead && !lds_wait(&cnt[C_REL], e_rel)) dead = true;                                                                       \
    }                                                                                                                                \
  } while (0)

  // ---- prefetch (issue only: nothing here waits) --------------------------------------------------------------------------------
#define PREFETCH_Q(L)                                                                                                                \
  do {                                                                                                                               \
    if (q_tilewave) {                                                                                                                \
      const uint4* wq = (L).Wqkv + ((size_t)q_tile * (PH >> 5) + q_slice * 8) * 64 + lane;                                           \
      _Pragma("unroll") for (int c = 0; c < 8; c++) P[c] = ldnt(wq + c * 64);                                                        \
      _Pragma("unroll") for (int b = 0; b < 2; b++) {                                                                                \
        const size_t ix = ((size_t)q_tile * PG + q_slice * 2 + b) * 64 + lane;                                                       \
        sA[b] = __half2float((L).Sqkv[ix]); zA[b] = (L).Zqkv[ix];                                                                    \
      }                                                                                                                              \
    }                                                                                                                                \
    nwa = *(const float4*)((L).attn_norm + i0); nwb = *(const float4*)((L).attn_norm + i0 + 4);                                      \
  } while (0)
#define PREFETCH_A(L, layer)                                                                                                         \
  do {                                                                                                                               \
    {                                                                                                                                \
      const uint4* wp = (L).Wo + ((size_t)(a_cs * 8 + wave) * (PH >> 5) + a_hq * 4) * 64 + lane;                                     \
      _Pragma("unroll") for (int c = 0; c < 4; c++) P[c] = ldnt(wp + c * 64);                                                        \
      const size_t ix = ((size_t)(a_cs * 8 + wave) * PG + a_hq) * 64 + lane;                                                         \
      sA[0] = __half2float((L).So[ix]); zA[0] = (L).Zo[ix];                                                                          \
    }                                                                                                                                \
    const unsigned short* kbp = (const unsigned short*)kv.k + (size_t)(layer) * kv.layer_stride;                                     \
    _Pragma("unroll") for (int i = 0; i < 8; i++)                                                                                    \
      P[8 + i] = *(const uint4*)(kbp + kv_row_off_t<PAGED>(kv, 0, a_kvh, min(wave * 32 + 4 * i + rsub, pmax)) + piece * 8);         \
  } while (0)
#define PREFETCH_M(L)                                                                                                                \
  do {                                                                                                                               \
    nwa = *(const float4*)((L).ffn_norm + i0); nwb = *(const float4*)((L).ffn_norm + i0 + 4);                                        \
    if (m_on) {                                                                                                                      \
      const uint4* wg_ = (L).Wgu + ((size_t)m_sl * (PH >> 5) + wave * 16) * 64 + lane;                                               \
      const uint4* wu_ = (L).Wgu + ((size_t)(NTI + m_sl) * (PH >> 5) + wave * 16) * 64 + lane;                                       \
      _Pragma("unroll") for (int c = 0; c < 8; c++) { P[c] = ldnt(wg_ + c * 64); P[8 + c] = ldnt(wu_ + c * 64); }                    \
      _Pragma("unroll") for (int b = 0; b < 4; b++) {                                                                                \
        const size_t ig = ((size_t)m_sl * PG + wave * 4 + b) * 64 + lane, iu = ((size_t)(NTI + m_sl) * PG + wave * 4 + b) * 64 + lane; \
        sA[b] = __half2float((L).Sgu[ig]); zA[b] = (L).Zgu[ig]; sB[b] = __half2float((L).Sgu[iu]); zB[b] = (L).Zgu[iu];              \
      }                                                                                                                              \
    }                                                                                                                                \
  } while (0)

  // ---- row update: h <- R(h + R(acc)) in LDS, exact sum of squares over the workgroup, x = R(w R(h rs)) -> this thread's plane words ------------------
  //      (HASACC = false: the first layer's row has nothing to add).  Leaves x's eight plane words in w[], the group sums in sp[], the scale in cs.
#define ROW_UPDATE(ACCPTR, HASACC, WRITE_OUT, DBG)                                                                                        \
  float x_[8]; unsigned w_[XQ_NP]; int sp_[XQ_NP]; float cs_; int4 g2w_;                                                             \
  {                                                                                                                                  \
    const float4 ha = *(const float4*)(S.hres + i0), hb = *(const float4*)(S.hres + i0 + 4);                                         \
    float v[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};                                                                   \
    if (HASACC) {                                                                                                                    \
      unsigned long long pv[8];                                                                                                      \
      _Pragma("unroll") for (int e = 0; e < 8; e++) pv[e] = ld_acc((ACCPTR) + i0 + e);                                               \
      _Pragma("unroll") for (int e = 0; e < 8; e++) v[e] = round_t<ACT>(v[e] + round_t<ACT>(fix2f((long long)pv[e], ACT)));          \
      *(float4*)(S.hres + i0) = make_float4(v[0], v[1], v[2], v[3]);                                                                 \
      *(float4*)(S.hres + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);                                                             \
    }                                                                                                                                \
    if (WRITE_OUT) { *(float4*)(a.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(a.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); } \
    double ssd = 0.0;                                                                                                                \
    _Pragma("unroll") for (int e = 0; e < 8; e += 2) ssd += (double)(v[e] * v[e]) + (double)(v[e + 1] * v[e + 1]);                   \
    ssd = wave_sum_d(ssd);                                                                                                           \
    if (lane == 0) { S.dred[wave] = ssd; lds_signal(&cnt[C_SS]); }                                                                   \
    e_ss += NW;                                                                                                                      \
    if (!dead && !lds_wait(&cnt[C_SS], e_ss)) dead = true;                                                                           \
    ssd = ((S.dred[0] + S.dred[1]) + (S.dred[2] + S.dred[3])) + ((S.dred[4] + S.dred[5]) + (S.dred[6] + S.dred[7]));                 \
    const float ss = (float)ssd;                                                                                                     \
    const float rs = rms_scale(ss, (float)PH, a.eps);                                                                           \
    if (a.stamps && wg == 0 && tid == 0) { a.stamps[24 + 64 * (DBG)] = __double_as_longlong(ssd); a.stamps[25 + 64 * (DBG)] = __float_as_int(ss); a.stamps[26 + 64 * (DBG)] = __float_as_int(rs); \
      for (int w8 = 0; w8 < 8; w8++) a.stamps[32 + 64 * (DBG) + w8] = __double_as_longlong(S.dred[w8]); }                                         \
    const float nwv[8] = {nwa.x, nwa.y, nwa.z, nwa.w, nwb.x, nwb.y, nwb.z, nwb.w};                                                   \
    float am = 0.f;                                                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 8; e++) { x_[e] = round_t<ACT>(nwv[e] * round_t<ACT>(v[e] * rs)); am = fmaxf(am, fabsf(x_[e])); } \
    am = grp_reduce<16, OpMax>(am);                                                                                                  \
    xq_split8(x_, am, w_, sp_, cs_);                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < XQ_NP; p++) sp_[p] = grp_reduce<16, OpAdd>(sp_[p]);                                        \
    g2w_ = xq_gpar_hi<16>(w_, sp_);                                                                                                  \
  }

  PREFETCH_Q(a.layers[0]);

  for (int l = 0; l < a.n_layers; l++) {
    // Every global address of the body is loop-invariant in its lane part (the layer only moves the scalar base), and LICM would hoist all of them out of
    // the layer loop -- a few hundred VGPRs of precomputed 64-bit addresses, i.e. 600 spilled registers.  The lane index is therefore made opaque once per
    // layer: everything derived from it is recomputed (a handful of VALU ops) instead of carried.
    int tid_l = threadIdx.x, lane_l;
    asm volatile("" : "+v"(tid_l));
    lane_l = tid_l & 63;
    const int tid = tid_l, lane = lane_l, oct = tid, i0 = oct * 8, piece = lane & 15, rsub = lane >> 4;
    PersistArgs a = a_in;
    asm volatile("" : "+s"(a.ring_m), "+s"(a.ring_q), "+s"(a.ring_o), "+s"(a.rope_cur), "+s"(a.h_out), "+s"(a.bar));
    const PLayer& L = a.layers[l];
    // =====================================================================================================================
    // Phase Q: h' = R(h + R(mlp_prev)) -> hres; RMSNorm; the workgroup's 256-k slice of the normalised row -> planes; 6 tiles x 256 k of q/k/v
    // =====================================================================================================================
    PSTAMP(0);
    {
      ROW_UPDATE(a.ring_m, l > 0, false, 0)
      PSTAMP(1);
      const int so = oct - q_osl;                                    // this thread's octet inside the slice (0..31) when it lies there
      if (so >= 0 && so < 32) {
        unsigned* plw = (unsigned*)S.xpl + ((so >> 2) * XQ_NP) * 4 + (so & 3);
#pragma unroll
        for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w_[p];
        if ((so & 15) == 0) { S.gpar[2 * (so >> 4)] = make_int4(__float_as_int(cs_), sp_[0], sp_[1], sp_[2]); S.gpar[2 * (so >> 4) + 1] = g2w_; }
      }
      if (wave == (q_osl >> 6) && lane == 0) lds_signal(&cnt[C_PL]);     // (lds_signal waits for the wave's LDS stores first)
      e_pl += 1;
      if (q_tilewave) {
        if (!dead && !lds_wait(&cnt[C_PL], e_pl)) dead = true;
        double y = 0.0;
        q4g_consume_at<0, 4, 16, true>(P, 0, 0, S.xpl, S.gpar, sA[0], zA[0], y);
        q4g_consume_at<4, 4, 16, true>(P, 4, 2, S.xpl, S.gpar, sA[1], zA[1], y);
        acc_add(a.ring_q + q_tile * 64 + lane, d2fix(y, ACT));
      }
      if (wave == 6 && lane < 16) acc_zero(a.ring_o + wg * 16 + lane);     // zero duty: o_proj accumulator (read in the previous M phase)
    }
    PSTAMP(2);
    vm_drain();
    PSTAMP(3);
    // (the control wave polls the barrier word with vector loads, which return in order behind everything it has in flight: it requests its share after the barrier)
    if (wave != CW) { PREFETCH_A(L, l); }
    GRID_BARRIER();
    if (wave == CW) { PREFETCH_A(L, l); }
    PSTAMP(4);

    // =====================================================================================================================
    // Phase A: q/k/v finish (+ RoPE), KV append, attention of head a_hq over the cache (exact sums, two passes), o_proj slab -> ring_o
    // =====================================================================================================================
    {
      const unsigned short* kb0 = (const unsigned short*)kv.k + (size_t)l * kv.layer_stride;
      const unsigned short* vb0 = (const unsigned short*)kv.v + (size_t)l * kv.layer_stride;
      __builtin_amdgcn_sched_barrier(0);
      uint4 V[8];     // V rows of the first chunk: requested now, used after the scores
#pragma unroll
      for (int i = 0; i < 8; i++) V[i] = *(const uint4*)(vb0 + kv_row_off_t<PAGED>(kv, 0, a_kvh, min(wave * 32 + 4 * i + rsub, pmax)) + piece * 8);
      if (wave < 3) {
        const int hh = wave, i = lane;
        const int base = hh == 0 ? a_hq * PHD : PNQ * PHD + a_kvh * PHD;
        const int vb = PNQ * PHD + PNKV * PHD + a_kvh * PHD + 2 * i;
        const int j0 = hh < 2 ? base + i : vb, j1 = hh < 2 ? base + i + 64 : vb + 1;
        const unsigned long long r0 = ld_acc(a.ring_q + j0), r1 = ld_acc(a.ring_q + j1);
        const float pc = a.rope_cur[i], ps = a.rope_cur[64 + i];
        const float px0 = round_t<ACT>(fix2f((long long)r0, ACT)), px1 = round_t<ACT>(fix2f((long long)r1, ACT));
        if (hh < 2) {
          const float y0 = round_t<ACT>(rope_lo(px0, px1, pc, ps)), y1 = round_t<ACT>(rope_hi(px0, px1, pc, ps));
          unsigned short* dst = (unsigned short*)(hh == 0 ? S.u.at.q2 : S.u.at.k2);
          const unsigned p0 = pack2<ACT>(y0, y1);
          dst[i] = (unsigned short)(p0 & 0xffffu);
          dst[i + 64] = (unsigned short)(p0 >> 16);
        } else {
          S.u.at.v2[i] = pack2<ACT>(px0, px1);
        }
        if (lane == 0) lds_signal(&cnt[C_QKV]);
      }
      e_qkv += 3;
      if (!dead && !lds_wait(&cnt[C_QKV], e_qkv)) dead = true;
      PSTAMP(5);
      if ((a_hq & 3) == 0 && a_cs == 0 && wave == 3) {   // KV append, once per kv head: 64 threads x 4-byte pairs
        size_t woff;
        if (PAGED) woff = kv_slot_off(kv, l, a_kvh, kv.slot ? kv.slot[0] : (kv.block_table[pos / kv.bs] * kv.bs + pos % kv.bs));
        else woff = kv_row_off_t<0>(kv, l, a_kvh, pos);
        ((unsigned*)((unsigned short*)kv.k + woff))[lane] = S.u.at.k2[lane];
        ((unsigned*)((unsigned short*)kv.v + woff))[lane] = S.u.at.v2[lane];
      }
      const float scale = div_rn(1.0f, sqrt_rn((float)PHD));
      const uint4 qq = ((const uint4*)S.u.at.q2)[piece];
      float qf[8];
      unpack2<ACT>(qq.x, qf[0], qf[1]); unpack2<ACT>(qq.y, qf[2], qf[3]); unpack2<ACT>(qq.z, qf[4], qf[5]); unpack2<ACT>(qq.w, qf[6], qf[7]);
      const bool single = len <= 256;
      float sc_[8];
      float Mw = -INFINITY;
#define P_SCORES(c0_)                                                                                                      \
      _Pragma("unroll") for (int i = 0; i < 8; i++) {                                                                      \
        const int p = (c0_) + wave * 32 + 4 * i + rsub;                                                                    \
        uint4 kk = P[8 + i];                                                                                               \
        if (p == pos) { kk = ((const uint4*)S.u.at.k2)[piece]; V[i] = ((const uint4*)S.u.at.v2)[piece]; }                  \
        float kf[8];                                                                                                       \
        unpack2<ACT>(kk.x, kf[0], kf[1]); unpack2<ACT>(kk.y, kf[2], kf[3]); unpack2<ACT>(kk.z, kf[4], kf[5]); unpack2<ACT>(kk.w, kf[6], kf[7]); \
        double d = 0.0;                                                                                                    \
        _Pragma("unroll") for (int e = 0; e < 8; e++) d = fma((double)kf[e], (double)qf[e], d);                            \
        d = grp_sum_d<16>(d);                                                                                              \
        sc_[i] = (p < len) ? (float)d * scale : -INFINITY;                                                                 \
      }
      for (int c0 = 0; c0 < len; c0 += 256) {
        if (c0 + wave * 32 >= len) continue;
        if (c0 > 0) {
#pragma unroll
          for (int i = 0; i < 8; i++) P[8 + i] = *(const uint4*)(kb0 + kv_row_off_t<PAGED>(kv, 0, a_kvh, min(c0 + wave * 32 + 4 * i + rsub, pmax)) + piece * 8);
        }
        P_SCORES(c0)
#pragma unroll
        for (int i = 0; i < 8; i++) Mw = fmaxf(Mw, sc_[i]);
      }
      Mw = wave_max(Mw);
      PSTAMP(6);
      if (lane == 0) { S.u.at.wred[wave] = Mw; lds_signal(&cnt[C_MAX]); }
      e_max += NW;
      if (!dead && !lds_wait(&cnt[C_MAX], e_max)) dead = true;
      float Mall = S.u.at.wred[0];
#pragma unroll
      for (int w = 1; w < NW; w++) Mall = fmaxf(Mall, S.u.at.wred[w]);
      double accv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      double lsum = 0.0;
      for (int c0 = 0; c0 < len; c0 += 256) {
        if (c0 + wave * 32 >= len) continue;
        if (!single) {
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const size_t off = kv_row_off_t<PAGED>(kv, 0, a_kvh, min(c0 + wave * 32 + 4 * i + rsub, pmax)) + piece * 8;
            P[8 + i] = *(const uint4*)(kb0 + off);
            V[i] = *(const uint4*)(vb0 + off);
          }
          P_SCORES(c0)
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float e = (sc_[i] == -INFINITY) ? 0.f : bz_expf(sc_[i] - Mall);
          const double ed = (double)e;
          lsum += ed;
          float v[8];
          unpack2<ACT>(V[i].x, v[0], v[1]); unpack2<ACT>(V[i].y, v[2], v[3]);
          unpack2<ACT>(V[i].z, v[4], v[5]); unpack2<ACT>(V[i].w, v[6], v[7]);
#pragma unroll
          for (int q = 0; q < 8; q++) accv[q] = fma(ed, (double)v[q], accv[q]);
        }
      }
#undef P_SCORES
      lsum = xrow32_d(xrow16_d(lsum));
#pragma unroll
      for (int q = 0; q < 8; q++) accv[q] = xrow32_d(xrow16_d(accv[q]));
      if (lane < 16) {
#pragma unroll
        for (int q = 0; q < 8; q++) S.u.at.pout[wave * PHD + piece * 8 + q] = accv[q];
      }
      if (lane == 0) { S.u.at.lred[wave] = lsum; lds_signal(&cnt[C_PV]); }
      PSTAMP(7);
      e_pv += NW;
      if (wave < 2) {
        if (!dead && !lds_wait(&cnt[C_PV], e_pv)) dead = true;
        double oc = 0.0, Lc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) { oc += S.u.at.pout[w * PHD + tid]; Lc += S.u.at.lred[w]; }
        S.u.at.outh[tid] = round_t<ACT>(div_rn((float)oc, (float)Lc));
        if (lane == 0) lds_signal(&cnt[C_OUT]);
      }
      e_out += 2;
      if (wave == 0) {     // the head output's planes: 16 threads x 8 values (one 128-k group)
        if (!dead && !lds_wait(&cnt[C_OUT], e_out)) dead = true;
        const bool on = tid < 16;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = on ? S.u.at.outh[tid * 8 + e] : 0.f;
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) am = fmaxf(am, fabsf(v[e]));
        am = grp_reduce<16, OpMax>(am);
        unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
        xq_split8(v, am, w, sp, cs);
#pragma unroll
        for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<16, OpAdd>(sp[p]);
        const int4 g2w = xq_gpar_hi<16>(w, sp);
        if (on) {
          unsigned* plw = (unsigned*)S.u.at.xpl + ((tid >> 2) * XQ_NP) * 4 + (tid & 3);
#pragma unroll
          for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w[p];
          if (tid == 0) { S.u.at.gpar[0] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]); S.u.at.gpar[1] = g2w; }
        }
        if (lane == 0) lds_signal(&cnt[C_QA]);
      }
      e_qa += 1;
      {
        if (!dead && !lds_wait(&cnt[C_QA], e_qa)) dead = true;
        PSTAMP(8);
        double y = 0.0;
        q4g_consume_at<0, 4, 16, true>(P, 0, 0, S.u.at.xpl, S.u.at.gpar, sA[0], zA[0], y);
        acc_add(a.ring_o + (a_cs * 8 + wave) * 64 + lane, d2fix(y, ACT));
      }
      if (wave == 6 && lane < 16 && l > 0) acc_zero(a.ring_m + wg * 16 + lane);   // zero duty: the previous MLP accumulator (read in phase Q)
    }
    PSTAMP(9);
    vm_drain();
    PSTAMP(10);
    if (wave != CW) { PREFETCH_M(L); }
    GRID_BARRIER();
    if (wave == CW) { PREFETCH_M(L); }
    PSTAMP(11);

    // =====================================================================================================================
    // Phase M: h'' = R(h' + R(o)) -> hres; RMSNorm; planes of the whole row; gate / up of 64 intermediate columns; SiLU * up; the 64-k slab of down
    // =====================================================================================================================
    {
      {
        ROW_UPDATE(a.ring_o, true, (l == a.n_layers - 1 && wg == 0), 1)
        unsigned* plw = (unsigned*)S.xpl + ((oct >> 2) * XQ_NP) * 4 + (oct & 3);
#pragma unroll
        for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w_[p];
        if ((lane & 15) == 0) { S.gpar[2 * (oct >> 4)] = make_int4(__float_as_int(cs_), sp_[0], sp_[1], sp_[2]); S.gpar[2 * (oct >> 4) + 1] = g2w_; }
        if (lane == 0) lds_signal(&cnt[C_PL]);
      }
      PSTAMP(12);
      e_pl += NW;
      if (m_on) {
        const int GD = I >> 7, gd = m_sl >> 1, tbeg = wave * 8;
        // The wave's stream in program order (scheduling barriers pin it: at most three groups' worth of weights -- 96 registers -- are live at a time):
        //   [groups 0, 1 prefetched]  load g2 | dots g0 | load g3 | dots g1 | load slab tiles 0..3 | dots g2 | load slab tiles 4..7 | dots g3
        const uint4* wg_ = L.Wgu + ((size_t)m_sl * (PH >> 5) + wave * 16) * 64 + lane;
        const uint4* wu_ = L.Wgu + ((size_t)(NTI + m_sl) * (PH >> 5) + wave * 16) * 64 + lane;
        uint4 G2[8], G3[8];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++) { G2[c] = ldnt(wg_ + (8 + c) * 64); G2[4 + c] = ldnt(wu_ + (8 + c) * 64); }
        __builtin_amdgcn_sched_barrier(0);
        if (!dead && !lds_wait(&cnt[C_PL], e_pl)) dead = true;
        double yg = 0.0, yu = 0.0;
        q4g_consume2_ab<0, 8, 16, true>(P, wave * 4, S.xpl, S.gpar, sA[0], zA[0], sB[0], zB[0], yg, yu);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++) { G3[c] = ldnt(wg_ + (12 + c) * 64); G3[4 + c] = ldnt(wu_ + (12 + c) * 64); }
        __builtin_amdgcn_sched_barrier(0);
        q4g_consume2_ab<4, 12, 16, true>(P, wave * 4 + 1, S.xpl, S.gpar, sA[1], zA[1], sB[1], zB[1], yg, yu);
        __builtin_amdgcn_sched_barrier(0);
        uint4 D[8][2];    // the down slab: 8 output tiles x the unit's two 32-k chunks
        float sd[8]; int zd[8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint4* wp = L.Wd + ((size_t)(tbeg + q) * (I >> 5) + 2 * m_sl) * 64 + lane;
          D[q][0] = ldnt(wp); D[q][1] = ldnt(wp + 64);
          const size_t si = ((size_t)(tbeg + q) * GD + gd) * 64 + lane;
          sd[q] = __half2float(L.Sd[si]); zd[q] = L.Zd[si];
        }
        __builtin_amdgcn_sched_barrier(0);
        q4g_consume2_ab<0, 4, 8, true>(G2, wave * 4 + 2, S.xpl, S.gpar, sA[2], zA[2], sB[2], zB[2], yg, yu);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4; q < 8; q++) {
          const uint4* wp = L.Wd + ((size_t)(tbeg + q) * (I >> 5) + 2 * m_sl) * 64 + lane;
          D[q][0] = ldnt(wp); D[q][1] = ldnt(wp + 64);
          const size_t si = ((size_t)(tbeg + q) * GD + gd) * 64 + lane;
          sd[q] = __half2float(L.Sd[si]); zd[q] = L.Zd[si];
        }
        __builtin_amdgcn_sched_barrier(0);
        q4g_consume2_ab<0, 4, 8, true>(G3, wave * 4 + 3, S.xpl, S.gpar, sA[3], zA[3], sB[3], zB[3], yg, yu);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(13);
        S.u.m.part[wave * 128 + lane] = yg;
        S.u.m.part[wave * 128 + 64 + lane] = yu;
        if (lane == 0) lds_signal(&cnt[C_PART]);
        e_part += NW;
        if (wave == 0) {
          if (!dead && !lds_wait(&cnt[C_PART], e_part)) dead = true;
          double tg = 0.0, tu = 0.0;
#pragma unroll
          for (int w2 = 0; w2 < NW; w2++) { tg += S.u.m.part[w2 * 128 + lane]; tu += S.u.m.part[w2 * 128 + 64 + lane]; }
          const float fg = (float)tg, fu = (float)tu;
          const float av = round_t<ACT>(round_t<ACT>(silu_f(round_t<ACT>(fg))) * round_t<ACT>(fu));
          const float am = wave_max(fabsf(av));
          const unsigned eb = (__float_as_uint(am) >> 23) & 255u;
          const bool live = eb >= 32u && eb < 255u;
          const float inv = live ? __uint_as_float((283u - eb) << 23) : 0.f;
          const float cs = live ? __uint_as_float((eb - 29u) << 23) : 0.f;
          const unsigned code = ((unsigned)(int)rintf(av * inv) + 0x88888888u) ^ 0x88888888u;
          const int lowfl = __builtin_amdgcn_ballot_w64((code & 0xFFu) != 0u) != 0ull;
          const int o = lane & 7, sh = 4 * (2 * (o & 3) + (o >> 2));
          int sp[XQ_NP];
#pragma unroll
          for (int p = 0; p < XQ_NP; p++) {
            const int nib = p < XQ_NM ? p + 2 : p - XQ_NM;
            const int wv = grp_reduce<8, OpOr>((int)(((code >> (4 * nib)) & 15u) << sh));
            if (o == 0) ((unsigned*)S.u.m.apl)[((lane >> 5) * XQ_NP + p) * 4 + ((lane >> 3) & 3)] = (unsigned)wv;
            int t = __builtin_amdgcn_sdot8(wv, 0x11111111, 0, false);
            t += dpp_get<DPP_ROR8>(t);
            const bz_u2_t r1 = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
            t = (int)r1.x + (int)r1.y;
            const bz_u2_t r2 = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
            sp[p] = (int)r2.x + (int)r2.y;
          }
          if (lane == 0) {
            S.u.m.apar[0] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
            S.u.m.apar[1] = make_int4(sp[3], sp[4], sp[5], xq_pack_low(sp[6], sp[7], lowfl));
            lds_signal(&cnt[C_TAIL]);
          }
        }
        e_tail += 1;
        if (!dead && !lds_wait(&cnt[C_TAIL], e_tail)) dead = true;
        PSTAMP(14);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          double y = 0.0;
          q4g_consume_at<0, 2, 2, false>(D[q], 0, 0, S.u.m.apl, S.u.m.apar, sd[q], zd[q], y);
          acc_add(a.ring_m + (tbeg + q) * 64 + lane, d2fix(y, ACT));
        }
      } else {
        e_part += NW; e_tail += 1;
      }
      if (wave == 6 && lane < 24) acc_zero(a.ring_q + wg * 24 + lane);     // zero duty: q/k/v accumulator (read in phase A)
    }
    PSTAMP(15);
    if (l + 1 < a.n_layers) {
      vm_drain();
      PSTAMP(16);
      if (wave != CW) { PREFETCH_Q(a.layers[l + 1]); }
      GRID_BARRIER();
      if (wave == CW) { PREFETCH_Q(a.layers[l + 1]); }
      PSTAMP(17);
    }
  }
#undef GRID_BARRIER
#undef PSTAMP
#undef PREFETCH_Q
#undef PREFETCH_A
#undef PREFETCH_M
#undef ROW_UPDATE
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
size_t bzk_persist_smem() { return sizeof(PersistLds); }
size_t bzk_persist_layer_bytes() { return sizeof(PLayer); }
size_t bzk_persist_bar_words() { return 11 * BAR_STRIDE; }

bool bzk_persist_shape_ok(int H, int I, int nq, int nkv, int hd, int act, int kv_dtype) {
  return H == PH && nq == PNQ && nkv == PNKV && hd == PHD && act == BZ_F16 && kv_dtype == BZ_F16 && I % 128 == 0 && I / 64 <= 256 && I >= 128;
}

int bzk_persist_fill_layer(void* host_entry, const LinearDev& qkv, const LinearDev& o, const LinearDev& gu, const LinearDev& dn, const float* attn_norm, const float* ffn_norm) {
  for (const LinearDev* L : {&qkv, &o, &gu, &dn})
    if (L->kind != LK_Q4G || L->perm || L->bias) BZ_FAIL(BZ_E_UNSUPPORTED, "persistent step: int4 group-quantised weights without act-order / bias only");
  PLayer p{};
  p.Wqkv = (const uint4*)qkv.w; p.Sqkv = (const __half*)qkv.scales; p.Zqkv = (const unsigned char*)qkv.zeros;
  p.Wo = (const uint4*)o.w; p.So = (const __half*)o.scales; p.Zo = (const unsigned char*)o.zeros;
  p.Wgu = (const uint4*)gu.w; p.Sgu = (const __half*)gu.scales; p.Zgu = (const unsigned char*)gu.zeros;
  p.Wd = (const uint4*)dn.w; p.Sd = (const __half*)dn.scales; p.Zd = (const unsigned char*)dn.zeros;
  p.attn_norm = attn_norm; p.ffn_norm = ffn_norm;
  memcpy(host_entry, &p, sizeof(p));
  return BZ_OK;
}

int bzk_llama_persist(hipStream_t s, const BzPersistLaunch& pl) {
  PersistArgs a{};
  a.layers = (const PLayer*)pl.layers; a.n_layers = pl.n_layers; a.h_in = pl.h_in; a.h_out = pl.h_out;
  a.ring_m = pl.ring_m; a.ring_q = pl.ring_q; a.ring_o = pl.ring_o; a.rope_cur = pl.rope_cur; a.pos = pl.pos; a.kv = pl.kv; a.bar = pl.bar; a.err_host = pl.err_host; a.eps = pl.eps; a.I = pl.I; a.stamps = pl.stamps;
  const size_t smem = sizeof(PersistLds);
  static bool attr_set = false;
  if (!attr_set) {
    BZ_HIP(hipFuncSetAttribute((const void*)k_llama_persist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    BZ_HIP(hipFuncSetAttribute((const void*)k_llama_persist<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  const double bytes = pl.algo_bytes;
  if (pl.kv.paged) BZ_LAUNCH("llama_persist<layers>", bytes, (k_llama_persist<1>), dim3(256), dim3(512), smem, s, a);
  else BZ_LAUNCH("llama_persist<layers>", bytes, (k_llama_persist<0>), dim3(256), dim3(512), smem, s, a);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
