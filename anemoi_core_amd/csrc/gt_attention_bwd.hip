// Graph-transformer edge attention, BACKWARD (materialised-E op boundary) for gfx950.
//
// Spec: the reference's two Triton passes _gt_bwd_dst_pass / _gt_bwd_src_pass and the op
// anemoi::graph_transformer_attention_backward (models/src/anemoi/models/triton/gt.py:182-376, 447-492).
// With s_e = <q_d, k_s + E_e> / sqrt(C), p_e = exp(s_e - m_d) (m = the forward's log-sum-exp), o_d = sum_e p_e (v_s + E_e):
//   D_d   = <dO_d, o_d>                      dS_e = p_e (<dO_d, v_s + E_e> - D_d)
//   dq_d  = sum_e dS_e (k_s + E_e) / sqrt(C)
//   dE_e  = dS_e q_d / sqrt(C) + p_e dO_d    dk_s = sum_{e from s} dS_e q_d / sqrt(C)    dv_s = sum_{e from s} p_e dO_d
//
// Not a port.  The reference recomputes s_e, p_e and dS_e in BOTH passes and writes dE from the source pass (rows
// scattered through the edge-id list).  Here:
//  * destination pass (one wave64 per destination, the forward's lane layout: VEC contiguous channels per lane, a head
//    = LPH adjacent lanes, DPP butterflies for the per-head sums): dq, dE - written once, in CSC order, whole contiguous
//    rows - and the two per-edge, per-head scalars p_e and dS_e / sqrt(C) into a small fp32 workspace [M, H] each;
//  * source pass (one wave64 per source over its out-edges via the reverse CSR): dk, dv from those scalars and the
//    gathered q_d / dO_d rows only - no dot products, no E traffic, no atomics, deterministic.
// HBM bytes (bf16): dst pass 2(3 N_dst D + 2 M D [k,v gathers, cached] + 2 M D [E in, dE out]) + 8 M H;
// src pass 2(2 N_src D) + gathers of q/dO + 8 M H.
#include "common.h"

namespace anemoi {

namespace {

constexpr int kBwdWaves = 4;

struct BwdArgs {
  const void *q, *k, *v, *e, *out, *d_out;
  int64_t ldq, ldk, ldv, lde, ldo, lddo;
  const float* lse;
  const int32_t *row, *colptr, *rowptr, *edge_ids, *edge_dst;
  void *dq, *dk, *dv, *de;
  int64_t lddq, lddk, lddv, ldde;
  float *p_ws, *ds_ws;
  int n_dst, n_src, H, C;
  hipStream_t stream;
  float drop_p = 0.f;  // attention dropout: the forward's probability and seed (the mask is re-derived, common.h)
  uint64_t drop_seed = 0;
};

// ---------------------------------------------------------------------------------------------- fast path
template <typename T, int VEC, int LPH>
__global__ __launch_bounds__(64 * kBwdWaves) void gt_attn_bwd_dst_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const T* __restrict__ e, int64_t lde, const T* __restrict__ out, int64_t ldo, const float* __restrict__ lse,
    const T* __restrict__ d_out, int64_t lddo, const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
    T* __restrict__ dq, int64_t lddq, T* __restrict__ de, int64_t ldde, float* __restrict__ p_ws, float* __restrict__ ds_ws,
    int n_dst, int H, float scale, float drop_p, uint64_t drop_seed) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kBwdWaves + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int c0 = lane * VEC;
  const int h = lane / LPH;
  const float inv_keep = 1.0f / (1.0f - drop_p);
  const int beg = colptr[d], end = colptr[d + 1];
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (end > beg) {
    float qv[VEC], gv[VEC], ov[VEC];
    load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
    load_vec<T, VEC>(d_out + (int64_t)d * lddo + c0, gv);
    load_vec<T, VEC>(out + (int64_t)d * ldo + c0, ov);
    float dd = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) dd = fmaf(gv[i], ov[i], dd);
    const float Dj = group_sum<LPH>(dd);
    const float m = lse[(int64_t)d * H + h];
    for (int ei = beg; ei < end; ++ei) {
      const int s = row[ei];
      float kv[VEC], vv[VEC], ev[VEC];
      load_vec<T, VEC>(k + (int64_t)s * ldk + c0, kv);
      load_vec<T, VEC>(v + (int64_t)s * ldv + c0, vv);
      load_vec<T, VEC>(e + (int64_t)ei * lde + c0, ev);
      float dot = 0.f, da = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        kv[i] += ev[i];
        dot = fmaf(qv[i], kv[i], dot);
        da = fmaf(gv[i], vv[i] + ev[i], da);
      }
      const float p = __expf(group_sum<LPH>(dot) * scale - m);
      // with dropout o_d = sum_e c_e p_e (v + E) (c_e = 0 or 1 / (1 - p)): d p_e = c_e <dO, v + E>, D_d = <dO, o_d> as before, and
      // the value-side weight (dE's second term, dv in the source pass) is c_e p_e
      const float pc = drop_p > 0.f ? p * attn_dropout_scale(drop_seed, ei, h, drop_p, inv_keep) : p;
      const float ds = (pc * group_sum<LPH>(da) - p * Dj) * scale;  // dS_e / sqrt(C)
      float dev[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = fmaf(ds, kv[i], acc[i]);
        dev[i] = fmaf(ds, qv[i], pc * gv[i]);
      }
      store_vec<T, VEC>(de + (int64_t)ei * ldde + c0, dev);
      if ((lane % LPH) == 0) {
        p_ws[(int64_t)ei * H + h] = pc;
        ds_ws[(int64_t)ei * H + h] = ds;
      }
    }
  }
  store_vec<T, VEC>(dq + (int64_t)d * lddq + c0, acc);
}

template <typename T, int VEC, int LPH>
__global__ __launch_bounds__(64 * kBwdWaves) void gt_attn_bwd_src_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ d_out, int64_t lddo, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ edge_ids, const int32_t* __restrict__ edge_dst, const float* __restrict__ p_ws,
    const float* __restrict__ ds_ws, T* __restrict__ dk, int64_t lddk, T* __restrict__ dv, int64_t lddv, int n_src, int H) {
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * kBwdWaves + (threadIdx.x >> 6));
  if (s >= n_src) return;
  const int c0 = lane * VEC;
  const int h = lane / LPH;
  float ak[VEC], av[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) ak[i] = av[i] = 0.f;
  const int beg = rowptr[s], end = rowptr[s + 1];
  // Out-edges in chunks of 64: edge ids and their destinations come from two coalesced loads (lane j holds edge j of the
  // chunk) and are broadcast with v_readlane; the q / dO rows of PF edges are in flight (clamped unconditional refills).
  using Raw = Vec<T, VEC>;
  constexpr int PF = 2;
  for (int chunk = beg; chunk < end; chunk += 64) {
    const int n = min(64, end - chunk);
    const int my_e = (lane < n) ? edge_ids[chunk + lane] : 0;
    const int my_d = (lane < n) ? edge_dst[my_e] : 0;
    Raw qb[PF], gb[PF];
    float pb[PF], sb[PF];
    auto fetch = [&](int j, Raw& qr, Raw& gr, float& pr, float& sr) {
      j = min(j, n - 1);
      const int ei = __builtin_amdgcn_readlane(my_e, j);
      const int d = __builtin_amdgcn_readlane(my_d, j);
      qr = *reinterpret_cast<const Raw*>(q + (int64_t)d * ldq + c0);
      gr = *reinterpret_cast<const Raw*>(d_out + (int64_t)d * lddo + c0);
      pr = p_ws[(int64_t)ei * H + h];
      sr = ds_ws[(int64_t)ei * H + h];
    };
#pragma unroll
    for (int st = 0; st < PF; ++st) fetch(st, qb[st], gb[st], pb[st], sb[st]);
    for (int j0 = 0; j0 < n; j0 += PF) {
#pragma unroll
      for (int st = 0; st < PF; ++st) {
        if (j0 + st < n) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            ak[j] = fmaf(sb[st], to_float(qb[st].v[j]), ak[j]);
            av[j] = fmaf(pb[st], to_float(gb[st].v[j]), av[j]);
          }
          fetch(j0 + st + PF, qb[st], gb[st], pb[st], sb[st]);
        }
      }
    }
  }
  store_vec<T, VEC>(dk + (int64_t)s * lddk + c0, ak);
  store_vec<T, VEC>(dv + (int64_t)s * lddv + c0, av);
}

// ---------------------------------------------------------------------------------------------- generic path
// any (H, C): one thread per (destination, head) / (source, head)
template <typename T>
__global__ void gt_attn_bwd_dst_generic_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                               const T* __restrict__ v, int64_t ldv, const T* __restrict__ e, int64_t lde,
                                               const T* __restrict__ out, int64_t ldo, const float* __restrict__ lse,
                                               const T* __restrict__ d_out, int64_t lddo, const int32_t* __restrict__ row,
                                               const int32_t* __restrict__ colptr, T* __restrict__ dq, int64_t lddq,
                                               T* __restrict__ de, int64_t ldde, float* __restrict__ p_ws,
                                               float* __restrict__ ds_ws, int n_dst, int H, int C, float scale, float drop_p,
                                               uint64_t drop_seed) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_dst * H) return;
  const int d = (int)(t / H), h = (int)(t % H);
  const int beg = colptr[d], end = colptr[d + 1];
  const float inv_keep = 1.0f / (1.0f - drop_p);
  const T* qp = q + (int64_t)d * ldq + h * C;
  const T* gp = d_out + (int64_t)d * lddo + h * C;
  const T* op = out + (int64_t)d * ldo + h * C;
  T* dqp = dq + (int64_t)d * lddq + h * C;
  for (int c = 0; c < C; ++c) dqp[c] = from_float<T>(0.f);
  if (end == beg) return;
  float Dj = 0.f;
  for (int c = 0; c < C; ++c) Dj = fmaf(to_float(gp[c]), to_float(op[c]), Dj);
  const float m = lse[(int64_t)d * H + h];
  // dq is accumulated in fp32 through a second sweep over the edges per channel block to stay register-only
  for (int ei = beg; ei < end; ++ei) {
    const int s = row[ei];
    const T* kp = k + (int64_t)s * ldk + h * C;
    const T* vp = v + (int64_t)s * ldv + h * C;
    const T* ep = e + (int64_t)ei * lde + h * C;
    float dot = 0.f, da = 0.f;
    for (int c = 0; c < C; ++c) {
      const float ee = to_float(ep[c]);
      dot = fmaf(to_float(qp[c]), to_float(kp[c]) + ee, dot);
      da = fmaf(to_float(gp[c]), to_float(vp[c]) + ee, da);
    }
    const float p = expf(dot * scale - m);
    const float pc = drop_p > 0.f ? p * attn_dropout_scale(drop_seed, ei, h, drop_p, inv_keep) : p;
    const float ds = (pc * da - p * Dj) * scale;
    p_ws[(int64_t)ei * H + h] = pc;
    ds_ws[(int64_t)ei * H + h] = ds;
    T* dep = de + (int64_t)ei * ldde + h * C;
    for (int c = 0; c < C; ++c) dep[c] = from_float<T>(fmaf(ds, to_float(qp[c]), pc * to_float(gp[c])));
  }
  for (int c = 0; c < C; ++c) {
    float a = 0.f;
    for (int ei = beg; ei < end; ++ei)
      a = fmaf(ds_ws[(int64_t)ei * H + h], to_float(k[(int64_t)row[ei] * ldk + h * C + c]) + to_float(e[(int64_t)ei * lde + h * C + c]), a);
    dqp[c] = from_float<T>(a);
  }
}

template <typename T>
__global__ void gt_attn_bwd_src_generic_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ d_out, int64_t lddo,
                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ edge_ids,
                                               const int32_t* __restrict__ edge_dst, const float* __restrict__ p_ws,
                                               const float* __restrict__ ds_ws, T* __restrict__ dk, int64_t lddk,
                                               T* __restrict__ dv, int64_t lddv, int n_src, int H, int C) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_src * H) return;
  const int s = (int)(t / H), h = (int)(t % H);
  const int beg = rowptr[s], end = rowptr[s + 1];
  for (int c = 0; c < C; ++c) {
    float ak = 0.f, av = 0.f;
    for (int i = beg; i < end; ++i) {
      const int ei = edge_ids[i];
      const int d = edge_dst[ei];
      ak = fmaf(ds_ws[(int64_t)ei * H + h], to_float(q[(int64_t)d * ldq + h * C + c]), ak);
      av = fmaf(p_ws[(int64_t)ei * H + h], to_float(d_out[(int64_t)d * lddo + h * C + c]), av);
    }
    dk[(int64_t)s * lddk + h * C + c] = from_float<T>(ak);
    dv[(int64_t)s * lddv + h * C + c] = from_float<T>(av);
  }
}

// ---------------------------------------------------------------------------------------------- dispatch
bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

template <typename T, int VEC, int LPH>
int launch_fast(const BwdArgs& a, float scale) {
  const dim3 block(64 * kBwdWaves);
  if (a.n_dst > 0) {
    hipLaunchKernelGGL((gt_attn_bwd_dst_kernel<T, VEC, LPH>), dim3((a.n_dst + kBwdWaves - 1) / kBwdWaves), block, 0, a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, (const T*)a.out,
                       a.ldo, a.lse, (const T*)a.d_out, a.lddo, a.row, a.colptr, (T*)a.dq, a.lddq, (T*)a.de, a.ldde, a.p_ws,
                       a.ds_ws, a.n_dst, a.H, scale, a.drop_p, a.drop_seed);
    const int rc = check_launch("gt_attn_bwd_dst_kernel");
    if (rc != ANEMOI_OK) return rc;
  }
  if (a.n_src > 0) {
    hipLaunchKernelGGL((gt_attn_bwd_src_kernel<T, VEC, LPH>), dim3((a.n_src + kBwdWaves - 1) / kBwdWaves), block, 0, a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.d_out, a.lddo, a.rowptr, a.edge_ids, a.edge_dst, a.p_ws, a.ds_ws,
                       (T*)a.dk, a.lddk, (T*)a.dv, a.lddv, a.n_src, a.H);
    return check_launch("gt_attn_bwd_src_kernel");
  }
  return ANEMOI_OK;
}

template <typename T, int VEC>
int launch_vec(const BwdArgs& a, float scale) {
  switch (a.C / VEC) {
    case 1: return launch_fast<T, VEC, 1>(a, scale);
    case 2: return launch_fast<T, VEC, 2>(a, scale);
    case 4: return launch_fast<T, VEC, 4>(a, scale);
    case 8: return launch_fast<T, VEC, 8>(a, scale);
    case 16: return launch_fast<T, VEC, 16>(a, scale);
    default: return 1;
  }
}

template <typename T>
int launch(const BwdArgs& a) {
  const int D = a.H * a.C;
  const float scale = 1.0f / sqrtf((float)a.C);
  const int64_t vb = 16 / (int64_t)sizeof(T);  // elements per 16 bytes
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool aligned = a.ldq % vb == 0 && a.ldk % vb == 0 && a.ldv % vb == 0 && a.lde % vb == 0 && a.ldo % vb == 0 &&
                       a.lddo % vb == 0 && a.lddq % vb == 0 && a.lddk % vb == 0 && a.lddv % vb == 0 && a.ldde % vb == 0 &&
                       al16(a.q) && al16(a.k) && al16(a.v) && al16(a.e) && al16(a.out) && al16(a.d_out) && al16(a.dq) &&
                       al16(a.dk) && al16(a.dv) && al16(a.de);
  int rc = 1;
  if (D % 64 == 0 && aligned) {
    const int vec = D / 64;
    if (a.C % vec == 0 && pow2(a.C / vec) && a.C / vec <= 16) {
      if (vec == 1) rc = launch_vec<T, 1>(a, scale);
      else if (vec == 2) rc = launch_vec<T, 2>(a, scale);
      else if (vec == 4) rc = launch_vec<T, 4>(a, scale);
      else if (vec == 8) rc = launch_vec<T, 8>(a, scale);
      else if (vec == 16) rc = launch_vec<T, 16>(a, scale);
    }
  }
  if (rc <= 0) return rc;
  const int64_t td = (int64_t)a.n_dst * a.H, ts = (int64_t)a.n_src * a.H;
  if (td > 0) {
    hipLaunchKernelGGL((gt_attn_bwd_dst_generic_kernel<T>), dim3((unsigned)((td + 127) / 128)), dim3(128), 0, a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, (const T*)a.out,
                       a.ldo, a.lse, (const T*)a.d_out, a.lddo, a.row, a.colptr, (T*)a.dq, a.lddq, (T*)a.de, a.ldde, a.p_ws,
                       a.ds_ws, a.n_dst, a.H, a.C, scale, a.drop_p, a.drop_seed);
    rc = check_launch("gt_attn_bwd_dst_generic_kernel");
    if (rc != ANEMOI_OK) return rc;
  }
  if (ts > 0) {
    hipLaunchKernelGGL((gt_attn_bwd_src_generic_kernel<T>), dim3((unsigned)((ts + 127) / 128)), dim3(128), 0, a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.d_out, a.lddo, a.rowptr, a.edge_ids, a.edge_dst, a.p_ws, a.ds_ws,
                       (T*)a.dk, a.lddk, (T*)a.dv, a.lddv, a.n_src, a.H, a.C);
    return check_launch("gt_attn_bwd_src_generic_kernel");
  }
  return ANEMOI_OK;
}


// ---------------------------------------------------------------------------------------------- fused lin_edge backward
// Backward of the op with lin_edge fused (E_e = W' f_e, W' = [W_e | b_e | 0] fp32 [D, FE_PAD], f_e = [edge_attr_e | 1 | 0]):
// E is never materialised here either.  With qw_h = W'_h^T q_d, gw_h = W'_h^T dO_d (per head h, FE_PAD values):
//   <q_d, E_e>_h = <f_e, qw_h>          <dO_d, E_e>_h = <f_e, gw_h>
//   dq_d   = sum_e dS_e k_s / sqrt(C) + W' (sum_e dS_e f_e / sqrt(C))            [per head sums sfS_h]
//   dW'    = sum_d dO_d (x) sfP_{d,h(c)} + q_d (x) sfS_{d,h(c)},   sfP_h = sum_e p_e f_e
//   df_e   = sum_h p_{e,h} gw_{d,h} + dS_{e,h} qw_{d,h} / sqrt(C)
// Destination pass: dq, the per-edge scalars for the source pass (unchanged), and the per-(destination, head) vectors
// sfP, sfS (and qw, gw when the edge attributes are trained) into fp32 workspaces; dW' and df are two small streaming kernels.
template <int VEC, int FE_PAD>
struct WLayoutB {
  static constexpr int kChunk = VEC * FE_PAD + 4;
  static constexpr int kFloats = 64 * kChunk;
};

template <typename T, int VEC, int LPH, int FE_PAD>
__global__ __launch_bounds__(64 * kBwdWaves) void gt_attn_bwd_dst_fused_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const float* __restrict__ feat, const float* __restrict__ w_packed, const T* __restrict__ out, int64_t ldo,
    const float* __restrict__ lse, const T* __restrict__ d_out, int64_t lddo, const int32_t* __restrict__ row,
    const int32_t* __restrict__ colptr, T* __restrict__ dq, int64_t lddq, float* __restrict__ p_ws, float* __restrict__ ds_ws,
    float* __restrict__ sf_ws, float* __restrict__ qg_ws, const T* __restrict__ addend, int64_t ldadd, T* __restrict__ d_addend,
    int64_t lddadd, int n_dst, int H, float scale) {
  // addend != NULL: `out` is the forward's result INCLUDING the addend (o = out - addend); d_addend != NULL: receives dO (the
  // gradient of the addend) - both save an elementwise kernel around the op.
  using L = WLayoutB<VEC, FE_PAD>;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // [64][kChunk], chunk layout [feature][channel]
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VEC;
  const int h = lane / LPH;
  for (int idx = threadIdx.x; idx < 64 * VEC * FE_PAD; idx += 64 * kBwdWaves) {
    const int c = idx / FE_PAD, f = idx % FE_PAD;
    w_lds[(c / VEC) * L::kChunk + f * VEC + (c % VEC)] = w_packed[idx];
  }
  __syncthreads();
  const float* wl = w_lds + lane * L::kChunk;
  const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * kBwdWaves + (threadIdx.x >> 6));
  const int nwaves = gridDim.x * kBwdWaves;
  for (int d = wave0; d < n_dst; d += nwaves) {
    asm volatile("" ::: "memory");  // keep W' in LDS (see the forward kernel)
    const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
    const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
    float qv[VEC], gv[VEC], ov[VEC], acc[VEC];
    load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
    load_vec<T, VEC>(d_out + (int64_t)d * lddo + c0, gv);
    load_vec<T, VEC>(out + (int64_t)d * ldo + c0, ov);
    if (addend != nullptr) {
      float av[VEC];
      load_vec<T, VEC>(addend + (int64_t)d * ldadd + c0, av);
#pragma unroll
      for (int i = 0; i < VEC; ++i) ov[i] -= av[i];
    }
    if (d_addend != nullptr) store_vec<T, VEC>(d_addend + (int64_t)d * lddadd + c0, gv);
    float dd = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      dd = fmaf(gv[i], ov[i], dd);
      acc[i] = 0.f;
    }
    const float Dj = group_sum<LPH>(dd);
    float qw[FE_PAD], gw[FE_PAD], sfP[FE_PAD], sfS[FE_PAD];
#pragma unroll
    for (int f = 0; f < FE_PAD; ++f) {
      float tq = 0.f, tg = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        tq = fmaf(qv[j], wl[f * VEC + j], tq);
        tg = fmaf(gv[j], wl[f * VEC + j], tg);
      }
      qw[f] = group_sum<LPH>(tq);
      gw[f] = group_sum<LPH>(tg);
      sfP[f] = sfS[f] = 0.f;
    }
    const float m = lse[(int64_t)d * H + h];
    // Edge loop as in the forward kernel: the source ids of a chunk come from ONE coalesced load, PF edges are in flight
    // (unconditional clamped refills keep the ring registers free of select code).
    using Raw = Vec<T, VEC>;
    constexpr int PF = 2;
    for (int chunk = beg; chunk < end; chunk += 64) {
      const int n = min(64, end - chunk);
      const int my_src = (lane < n) ? row[chunk + lane] : 0;
      Raw kb[PF], vb[PF];
      float fb[PF][FE_PAD];
      auto fetch = [&](int j, Raw& kr, Raw& vr, float (&fr)[FE_PAD]) {
        j = min(j, n - 1);
        const int s = __builtin_amdgcn_readlane(my_src, j);
        kr = *reinterpret_cast<const Raw*>(k + (int64_t)s * ldk + c0);
        vr = *reinterpret_cast<const Raw*>(v + (int64_t)s * ldv + c0);
        const float* a = feat + (int64_t)(chunk + j) * FE_PAD;  // wave-uniform address -> scalar loads
#pragma unroll
        for (int f = 0; f < FE_PAD; ++f) fr[f] = a[f];
      };
#pragma unroll
      for (int st = 0; st < PF; ++st) fetch(st, kb[st], vb[st], fb[st]);
      for (int j0 = 0; j0 < n; j0 += PF) {
#pragma unroll
        for (int st = 0; st < PF; ++st) {
          const int j = j0 + st;
          if (j < n) {
            float dot = 0.f, da = 0.f;
            float kv[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              kv[i] = to_float(kb[st].v[i]);
              dot = fmaf(qv[i], kv[i], dot);
              da = fmaf(gv[i], to_float(vb[st].v[i]), da);
            }
            dot = group_sum<LPH>(dot);
            da = group_sum<LPH>(da);
#pragma unroll
            for (int f = 0; f < FE_PAD; ++f) {
              dot = fmaf(fb[st][f], qw[f], dot);
              da = fmaf(fb[st][f], gw[f], da);
            }
            const float p = __expf(dot * scale - m);
            const float ds = p * (da - Dj) * scale;  // dS_e / sqrt(C)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(ds, kv[i], acc[i]);
#pragma unroll
            for (int f = 0; f < FE_PAD; ++f) {
              sfP[f] = fmaf(p, fb[st][f], sfP[f]);
              sfS[f] = fmaf(ds, fb[st][f], sfS[f]);
            }
            if ((lane % LPH) == 0) {
              p_ws[(int64_t)(chunk + j) * H + h] = p;
              ds_ws[(int64_t)(chunk + j) * H + h] = ds;
            }
            fetch(j + PF, kb[st], vb[st], fb[st]);
          }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FE_PAD; ++f)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wl[f * VEC + j], sfS[f], acc[j]);
    store_vec<T, VEC>(dq + (int64_t)d * lddq + c0, acc);
    if ((lane % LPH) == 0) {  // every (destination, head) slot is written, also with no in-edges (zeros); 16-byte stores
      float4* sp = reinterpret_cast<float4*>(sf_ws + ((int64_t)d * H + h) * 2 * FE_PAD);
#pragma unroll
      for (int t = 0; t < FE_PAD / 4; ++t) {
        sp[t] = float4{sfP[4 * t], sfP[4 * t + 1], sfP[4 * t + 2], sfP[4 * t + 3]};
        sp[FE_PAD / 4 + t] = float4{sfS[4 * t], sfS[4 * t + 1], sfS[4 * t + 2], sfS[4 * t + 3]};
      }
      if (qg_ws != nullptr) {  // qw as it is: the 1/sqrt(C) is already inside dS
        float4* gp = reinterpret_cast<float4*>(qg_ws + ((int64_t)d * H + h) * 2 * FE_PAD);
#pragma unroll
        for (int t = 0; t < FE_PAD / 4; ++t) {
          gp[t] = float4{qw[4 * t], qw[4 * t + 1], qw[4 * t + 2], qw[4 * t + 3]};
          gp[FE_PAD / 4 + t] = float4{gw[4 * t], gw[4 * t + 1], gw[4 * t + 2], gw[4 * t + 3]};
        }
      }
    }
  }
}

// dW'[c][f] partial sums: persistent waves over the destinations, fixed-order in-block sum, one fp32 row per block.
constexpr int kWgradBlocks = 256;
template <typename T, int VEC, int LPH, int FE_PAD>
__global__ __launch_bounds__(64 * kBwdWaves) void edge_weight_grad_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ d_out,
                                                                          int64_t lddo, const float* __restrict__ sf_ws,
                                                                          float* __restrict__ part, int n_dst, int H) {
  extern __shared__ float block_sum[];  // [FE_PAD][64 * VEC]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * VEC, h = lane / LPH;
  float dw[FE_PAD][VEC];
#pragma unroll
  for (int f = 0; f < FE_PAD; ++f)
#pragma unroll
    for (int j = 0; j < VEC; ++j) dw[f][j] = 0.f;
  // rows d, d + stride, ...: the loads of the next row are issued before the current one is accumulated
  using Raw = Vec<T, VEC>;
  const int stride = gridDim.x * kBwdWaves;
  int d = blockIdx.x * kBwdWaves + wave;
  Raw q_n{}, g_n{};
  float4 s_n[FE_PAD / 2];
  auto fetch = [&](int dd) {
    dd = min(dd, n_dst - 1);  // unconditional (clamped) refill
    q_n = *reinterpret_cast<const Raw*>(q + (int64_t)dd * ldq + c0);
    g_n = *reinterpret_cast<const Raw*>(d_out + (int64_t)dd * lddo + c0);
    const float4* sp = reinterpret_cast<const float4*>(sf_ws + ((int64_t)dd * H + h) * 2 * FE_PAD);  // FE_PAD % 4 == 0
#pragma unroll
    for (int t = 0; t < FE_PAD / 2; ++t) s_n[t] = sp[t];
  };
  if (d < n_dst) fetch(d);
  for (; d < n_dst; d += stride) {
    float qv[VEC], gv[VEC], sv[2 * FE_PAD];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      qv[j] = to_float(q_n.v[j]);
      gv[j] = to_float(g_n.v[j]);
    }
#pragma unroll
    for (int t = 0; t < FE_PAD / 2; ++t) {
      sv[4 * t] = s_n[t].x;
      sv[4 * t + 1] = s_n[t].y;
      sv[4 * t + 2] = s_n[t].z;
      sv[4 * t + 3] = s_n[t].w;
    }
    fetch(d + stride);
#pragma unroll
    for (int f = 0; f < FE_PAD; ++f) {
      const float a = sv[f], b = sv[FE_PAD + f];
#pragma unroll
      for (int j = 0; j < VEC; ++j) dw[f][j] = fmaf(gv[j], a, fmaf(qv[j], b, dw[f][j]));
    }
  }
  for (int kq = 0; kq < kBwdWaves; ++kq) {  // fixed order
    if (wave == kq) {
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float* dst = block_sum + f * (64 * VEC) + c0 + j;  // [feature][channel]: a lane's channels are contiguous, no bank conflicts
          *dst = kq == 0 ? dw[f][j] : *dst + dw[f][j];
        }
    }
    __syncthreads();
  }
  float* prow = part + (int64_t)blockIdx.x * 64 * VEC * FE_PAD;
  for (int i = threadIdx.x; i < 64 * VEC * FE_PAD; i += 64 * kBwdWaves) prow[i] = block_sum[i];
}

// out[c][f] = sum_b part[b][f][c] (the partial rows are [feature][channel], the result [channel][feature] like w_packed)
__global__ __launch_bounds__(1024) void sum_partial_rows_kernel(const float* __restrict__ part, int n_part, int n, int D, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int b = grp;
    for (; b + 16 < n_part; b += 32) {  // two independent chains of loads
      s0 += part[(int64_t)b * n + i];
      s1 += part[(int64_t)(b + 16) * n + i];
    }
    if (b < n_part) s0 += part[(int64_t)b * n + i];
  }
  red[grp][cl] = s0 + s1;
  __syncthreads();
#pragma unroll
  for (int step = 8; step >= 1; step >>= 1) {
    if (grp < step) red[grp][cl] += red[grp + step][cl];
    __syncthreads();
  }
  if (grp == 0 && i < n) out[(i % D) * (n / D) + i / D] = red[0][cl];
}

// df[e][f] = sum_h p[e,h] gw[d,h,f] + ds[e,h] qw[d,h,f]: 16 lanes per edge (lane = feature), edges sorted by destination.
__global__ void edge_feat_grad_kernel(const float* __restrict__ p_ws, const float* __restrict__ ds_ws, const float* __restrict__ qg_ws,
                                      const int32_t* __restrict__ edge_dst, float* __restrict__ d_feat, int n_edges, int H, int fe_pad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t >> 4;
  const int f = (int)(t & 15);
  if (e >= n_edges || f >= fe_pad) return;
  const int d = edge_dst[e];
  float s0 = 0.f, s1 = 0.f;  // two independent chains; the loads of several heads are in flight together
  int hh = 0;
#pragma unroll 4
  for (; hh + 1 < H; hh += 2) {
    const float* g0 = qg_ws + ((int64_t)d * H + hh) * 2 * fe_pad;
    const float* g1 = g0 + 2 * fe_pad;
    s0 = fmaf(p_ws[e * H + hh], g0[fe_pad + f], fmaf(ds_ws[e * H + hh], g0[f], s0));
    s1 = fmaf(p_ws[e * H + hh + 1], g1[fe_pad + f], fmaf(ds_ws[e * H + hh + 1], g1[f], s1));
  }
  if (hh < H) {
    const float* g0 = qg_ws + ((int64_t)d * H + hh) * 2 * fe_pad;
    s0 = fmaf(p_ws[e * H + hh], g0[fe_pad + f], fmaf(ds_ws[e * H + hh], g0[f], s0));
  }
  d_feat[e * fe_pad + f] = s0 + s1;
}

struct FusedBwdArgs {
  BwdArgs b;
  const float *feat, *w_packed;
  int fe_pad;
  float *d_w_packed, *d_feat, *sf_ws, *qg_ws, *part_ws;
  int n_edges;
  const void* addend;
  int64_t ldadd;
  void* d_addend;
  int64_t lddadd;
};

template <typename T, int VEC, int LPH, int FE_PAD>
int launch_fused_cfg(const FusedBwdArgs& f, float scale) {
  using L = WLayoutB<VEC, FE_PAD>;
  const BwdArgs& a = f.b;
  const dim3 block(64 * kBwdWaves);
  if (L::kFloats * sizeof(float) > 64 * 1024) return 1;
  if (a.n_dst > 0) {
    int blocks = (a.n_dst + kBwdWaves - 1) / kBwdWaves;
    blocks = blocks < 256 * 6 ? blocks : 256 * 6;
    hipLaunchKernelGGL((gt_attn_bwd_dst_fused_kernel<T, VEC, LPH, FE_PAD>), dim3(blocks), block, L::kFloats * sizeof(float), a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, f.feat, f.w_packed, (const T*)a.out, a.ldo,
                       a.lse, (const T*)a.d_out, a.lddo, a.row, a.colptr, (T*)a.dq, a.lddq, a.p_ws, a.ds_ws, f.sf_ws, f.qg_ws,
                       (const T*)f.addend, f.ldadd, (T*)f.d_addend, f.lddadd, a.n_dst, a.H, scale);
    int rc = check_launch("gt_attn_bwd_dst_fused_kernel");
    if (rc != ANEMOI_OK) return rc;
  }
  const int nw = 64 * VEC * FE_PAD;
  int wb = (a.n_dst + kBwdWaves - 1) / kBwdWaves;
  wb = wb < kWgradBlocks ? wb : kWgradBlocks;
  if (wb > 0) {
    hipLaunchKernelGGL((edge_weight_grad_kernel<T, VEC, LPH, FE_PAD>), dim3(wb), block, nw * sizeof(float), a.stream, (const T*)a.q, a.ldq,
                       (const T*)a.d_out, a.lddo, f.sf_ws, f.part_ws, a.n_dst, a.H);
    int rc = check_launch("edge_weight_grad_kernel");
    if (rc != ANEMOI_OK) return rc;
  }
  hipLaunchKernelGGL(sum_partial_rows_kernel, dim3((nw + 63) / 64), dim3(1024), 0, a.stream, f.part_ws, wb, nw, 64 * VEC, f.d_w_packed);
  int rc = check_launch("sum_partial_rows_kernel");
  if (rc != ANEMOI_OK) return rc;
  if (f.d_feat != nullptr && f.n_edges > 0) {
    const int64_t threads = (int64_t)f.n_edges * 16;
    hipLaunchKernelGGL(edge_feat_grad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, a.stream, a.p_ws, a.ds_ws, f.qg_ws,
                       a.edge_dst, f.d_feat, f.n_edges, a.H, f.fe_pad);
    rc = check_launch("edge_feat_grad_kernel");
    if (rc != ANEMOI_OK) return rc;
  }
  if (a.n_src > 0) {
    hipLaunchKernelGGL((gt_attn_bwd_src_kernel<T, VEC, LPH>), dim3((a.n_src + kBwdWaves - 1) / kBwdWaves), block, 0, a.stream,
                       (const T*)a.q, a.ldq, (const T*)a.d_out, a.lddo, a.rowptr, a.edge_ids, a.edge_dst, a.p_ws, a.ds_ws,
                       (T*)a.dk, a.lddk, (T*)a.dv, a.lddv, a.n_src, a.H);
    return check_launch("gt_attn_bwd_src_kernel");
  }
  return ANEMOI_OK;
}

template <typename T, int VEC, int LPH>
int launch_fused_fe(const FusedBwdArgs& f, float scale) {
  switch (f.fe_pad) {
    case 4: return launch_fused_cfg<T, VEC, LPH, 4>(f, scale);
    case 8: return launch_fused_cfg<T, VEC, LPH, 8>(f, scale);
    case 12: return launch_fused_cfg<T, VEC, LPH, 12>(f, scale);
    case 16: return launch_fused_cfg<T, VEC, LPH, 16>(f, scale);
    default: return 1;
  }
}

// The fused backward covers the shapes the production configs use (H*C = 64 * VEC with VEC = 8 for 16-bit / 4 or 8 for fp32
// models of 256 / 512 channels, and the 64-channel test models); anything else reports ANEMOI_E_UNSUPPORTED and the caller
// trains through the materialised-E op.
template <typename T>
int launch_fused(const FusedBwdArgs& f) {
  const BwdArgs& a = f.b;
  const int D = a.H * a.C;
  const float scale = 1.0f / sqrtf((float)a.C);
  const int64_t vb = 16 / (int64_t)sizeof(T);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool aligned = a.ldq % vb == 0 && a.ldk % vb == 0 && a.ldv % vb == 0 && a.ldo % vb == 0 && a.lddo % vb == 0 && a.lddq % vb == 0 &&
                       a.lddk % vb == 0 && a.lddv % vb == 0 && al16(a.q) && al16(a.k) && al16(a.v) && al16(a.out) && al16(a.d_out) &&
                       al16(a.dq) && al16(a.dk) && al16(a.dv) && al16(f.addend) && al16(f.d_addend) && f.ldadd % vb == 0 && f.lddadd % vb == 0;
  int rc = 1;
  if (D % 64 == 0 && aligned) {
    const int vec = D / 64, lph = a.C % vec == 0 ? a.C / vec : 0;
    if (vec == 8 && lph == 4) rc = launch_fused_fe<T, 8, 4>(f, scale);        // 512 channels, C = 32
    else if (vec == 8 && lph == 8) rc = launch_fused_fe<T, 8, 8>(f, scale);   // 512 channels, C = 64
    else if (vec == 4 && lph == 8) rc = launch_fused_fe<T, 4, 8>(f, scale);   // 256 channels, C = 32
    else if (vec == 1 && lph == 16) rc = launch_fused_fe<T, 1, 16>(f, scale); // 64 channels, C = 16 (test models)
    else if (vec == 1 && lph == 8) rc = launch_fused_fe<T, 1, 8>(f, scale);   // 64 channels, C = 8
  }
  if (rc > 0) {
    set_error("gt_attention_fused_edge_bwd: shape H=%d C=%d fe_pad=%d is not covered by the fused backward", a.H, a.C, f.fe_pad);
    return ANEMOI_E_UNSUPPORTED;
  }
  return rc;
}

}  // namespace
}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_attention_dropout_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const void* out, int64_t ldo, const float* lse,
                                       const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                                       const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst, void* dq,
                                       int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* de, int64_t ldde,
                                       float* p_ws, float* ds_ws, int32_t n_dst, int32_t n_src, int32_t n_edges, int32_t H,
                                       int32_t C, float drop_p, uint64_t drop_seed, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(drop_p >= 0.f && drop_p < 1.f && H < 65536, "gt_attention_dropout_bwd: dropout probability %g outside [0, 1) or H=%d too large", (double)drop_p, H);
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && n_edges >= 0 && H > 0 && C > 0, "gt_attention_dropout_bwd: bad sizes n_dst=%d n_src=%d M=%d H=%d C=%d",
                 n_dst, n_src, n_edges, H, C);
  if (n_dst == 0 && n_src == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(colptr && rowptr && dq && dk && dv, "gt_attention_dropout_bwd: null colptr/rowptr/dq/dk/dv");
  ANEMOI_REQUIRE(n_dst == 0 || (q && out && lse && d_out), "gt_attention_dropout_bwd: null q/out/lse/d_out");
  ANEMOI_REQUIRE(n_edges == 0 || (k && v && e && row && edge_ids && edge_dst && de && p_ws && ds_ws),
                 "gt_attention_dropout_bwd: null k/v/e/row/edge_ids/edge_dst/de/workspace with %d edges", n_edges);
  const int64_t D = (int64_t)H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && lde >= D && ldo >= D && lddo >= D && lddq >= D && lddk >= D && lddv >= D && ldde >= D,
                 "gt_attention_dropout_bwd: leading dimension smaller than H*C=%lld", (long long)D);
  BwdArgs a{q, k, v, e, out, d_out, ldq, ldk, ldv, lde, ldo, lddo, lse, row, colptr, rowptr, edge_ids, edge_dst,
            dq, dk, dv, de, lddq, lddk, lddv, ldde, p_ws, ds_ws, n_dst, n_src, H, C, as_stream(stream)};
  a.drop_p = drop_p;
  a.drop_seed = drop_seed;
  switch (dtype) {
    case ANEMOI_F32: return launch<float>(a);
    case ANEMOI_BF16: return launch<bf16_t>(a);
    case ANEMOI_F16: return launch<f16_t>(a);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_gt_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const void* out, int64_t ldo, const float* lse,
                                       const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                                       const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst, void* dq,
                                       int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* de, int64_t ldde,
                                       float* p_ws, float* ds_ws, int32_t n_dst, int32_t n_src, int32_t n_edges, int32_t H,
                                       int32_t C, anemoi_dtype_t dtype, void* stream) {
  return anemoi_gt_attention_dropout_bwd(q, ldq, k, ldk, v, ldv, e, lde, out, ldo, lse, d_out, lddo, row, colptr, rowptr, edge_ids, edge_dst, dq, lddq,
                                         dk, lddk, dv, lddv, de, ldde, p_ws, ds_ws, n_dst, n_src, n_edges, H, C, 0.f, 0, dtype, stream);
}

extern "C" int64_t anemoi_gt_attention_fused_edge_bwd_partial_floats(int32_t H, int32_t C, int32_t fe_pad) {
  return (int64_t)kWgradBlocks * H * C * fe_pad;
}

extern "C" int anemoi_gt_attention_fused_edge_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                                  const float* edge_feat, int32_t fe_pad, const float* w_packed, const void* out,
                                                  int64_t ldo, const float* lse, const void* d_out, int64_t lddo, const int32_t* row,
                                                  const int32_t* colptr, const int32_t* rowptr, const int32_t* edge_ids,
                                                  const int32_t* edge_dst, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                                                  int64_t lddv, float* d_w_packed, float* d_edge_feat, float* p_ws, float* ds_ws,
                                                  float* sf_ws, float* qg_ws, float* part_ws, const void* addend, int64_t ldadd,
                                                  void* d_addend, int64_t lddadd, int32_t n_dst, int32_t n_src, int32_t n_edges,
                                                  int32_t H, int32_t C, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_dst > 0 && n_src > 0 && n_edges >= 0 && H > 0 && C > 0 && fe_pad > 0 && fe_pad % 4 == 0,
                 "gt_attention_fused_edge_bwd: bad sizes n_dst=%d n_src=%d M=%d H=%d C=%d fe_pad=%d", n_dst, n_src, n_edges, H, C, fe_pad);
  ANEMOI_REQUIRE(q && k && v && out && lse && d_out && colptr && rowptr && dq && dk && dv && w_packed && d_w_packed && sf_ws && part_ws,
                 "gt_attention_fused_edge_bwd: null pointer");
  ANEMOI_REQUIRE(n_edges == 0 || (edge_feat && row && edge_ids && edge_dst && p_ws && ds_ws), "gt_attention_fused_edge_bwd: null edge data");
  ANEMOI_REQUIRE(d_edge_feat == nullptr || qg_ws != nullptr, "gt_attention_fused_edge_bwd: d_edge_feat needs the qg workspace");
  const int64_t D = (int64_t)H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D && lddo >= D && lddq >= D && lddk >= D && lddv >= D,
                 "gt_attention_fused_edge_bwd: leading dimension smaller than H*C=%lld", (long long)D);
  FusedBwdArgs f{{q, k, v, nullptr, out, d_out, ldq, ldk, ldv, 0, ldo, lddo, lse, row, colptr, rowptr, edge_ids, edge_dst,
                  dq, dk, dv, nullptr, lddq, lddk, lddv, 0, p_ws, ds_ws, n_dst, n_src, H, C, as_stream(stream)},
                 edge_feat, w_packed, fe_pad, d_w_packed, d_edge_feat, sf_ws, d_edge_feat ? qg_ws : nullptr, part_ws, n_edges,
                 addend, ldadd, d_addend, lddadd};
  ANEMOI_REQUIRE((addend == nullptr || ldadd >= D) && (d_addend == nullptr || lddadd >= D), "gt_attention_fused_edge_bwd: addend stride");
  switch (dtype) {
    case ANEMOI_F32: return launch_fused<float>(f);
    case ANEMOI_BF16: return launch_fused<bf16_t>(f);
    case ANEMOI_F16: return launch_fused<f16_t>(f);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}
