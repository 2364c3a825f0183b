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
};

// ---------------------------------------------------------------------------------------------- fast path
template <typename T, int VEC, int LPH>
__global__ __launch_bounds__(64 * kBwdWaves) void gt_attn_bwd_dst_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const T* __restrict__ e, int64_t lde, const T* __restrict__ out, int64_t ldo, const float* __restrict__ lse,
    const T* __restrict__ d_out, int64_t lddo, const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
    T* __restrict__ dq, int64_t lddq, T* __restrict__ de, int64_t ldde, float* __restrict__ p_ws, float* __restrict__ ds_ws,
    int n_dst, int H, float scale) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kBwdWaves + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int c0 = lane * VEC;
  const int h = lane / LPH;
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
      const float ds = p * (group_sum<LPH>(da) - Dj) * scale;  // dS_e / sqrt(C)
      float dev[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = fmaf(ds, kv[i], acc[i]);
        dev[i] = fmaf(ds, qv[i], p * gv[i]);
      }
      store_vec<T, VEC>(de + (int64_t)ei * ldde + c0, dev);
      if ((lane % LPH) == 0) {
        p_ws[(int64_t)ei * H + h] = p;
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
  for (int i = beg; i < end; ++i) {
    const int ei = edge_ids[i];
    const int d = edge_dst[ei];
    float qv[VEC], gv[VEC];
    load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
    load_vec<T, VEC>(d_out + (int64_t)d * lddo + c0, gv);
    const float p = p_ws[(int64_t)ei * H + h], ds = ds_ws[(int64_t)ei * H + h];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      ak[j] = fmaf(ds, qv[j], ak[j]);
      av[j] = fmaf(p, gv[j], av[j]);
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
                                               float* __restrict__ ds_ws, int n_dst, int H, int C, float scale) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_dst * H) return;
  const int d = (int)(t / H), h = (int)(t % H);
  const int beg = colptr[d], end = colptr[d + 1];
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
    const float ds = p * (da - Dj) * scale;
    p_ws[(int64_t)ei * H + h] = p;
    ds_ws[(int64_t)ei * H + h] = ds;
    T* dep = de + (int64_t)ei * ldde + h * C;
    for (int c = 0; c < C; ++c) dep[c] = from_float<T>(fmaf(ds, to_float(qp[c]), p * to_float(gp[c])));
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
                       a.ds_ws, a.n_dst, a.H, scale);
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
                       a.ds_ws, a.n_dst, a.H, a.C, scale);
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

}  // namespace
}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const void* out, int64_t ldo, const float* lse,
                                       const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                                       const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst, void* dq,
                                       int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* de, int64_t ldde,
                                       float* p_ws, float* ds_ws, int32_t n_dst, int32_t n_src, int32_t n_edges, int32_t H,
                                       int32_t C, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && n_edges >= 0 && H > 0 && C > 0, "gt_attention_bwd: bad sizes n_dst=%d n_src=%d M=%d H=%d C=%d",
                 n_dst, n_src, n_edges, H, C);
  if (n_dst == 0 && n_src == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(colptr && rowptr && dq && dk && dv, "gt_attention_bwd: null colptr/rowptr/dq/dk/dv");
  ANEMOI_REQUIRE(n_dst == 0 || (q && out && lse && d_out), "gt_attention_bwd: null q/out/lse/d_out");
  ANEMOI_REQUIRE(n_edges == 0 || (k && v && e && row && edge_ids && edge_dst && de && p_ws && ds_ws),
                 "gt_attention_bwd: null k/v/e/row/edge_ids/edge_dst/de/workspace with %d edges", n_edges);
  const int64_t D = (int64_t)H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && lde >= D && ldo >= D && lddo >= D && lddq >= D && lddk >= D && lddv >= D && ldde >= D,
                 "gt_attention_bwd: leading dimension smaller than H*C=%lld", (long long)D);
  BwdArgs a{q, k, v, e, out, d_out, ldq, ldk, ldv, lde, ldo, lddo, lse, row, colptr, rowptr, edge_ids, edge_dst,
            dq, dk, dv, de, lddq, lddk, lddv, ldde, p_ws, ds_ws, n_dst, n_src, H, C, as_stream(stream)};
  switch (dtype) {
    case ANEMOI_F32: return launch<float>(a);
    case ANEMOI_BF16: return launch<bf16_t>(a);
    case ANEMOI_F16: return launch<f16_t>(a);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}
