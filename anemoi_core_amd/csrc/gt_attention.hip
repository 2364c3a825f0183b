// Fused graph-transformer edge attention (forward) for gfx950.
//
// Spec: the reference's _gt_fwd Triton kernel (models/src/anemoi/models/triton/gt.py:81-179) and
// GraphTransformerConv (models/src/anemoi/models/layers/conv.py:84-147).  Not a port: the Triton kernel
// runs one program per destination with [H,C] register tiles; here one 64-lane wavefront owns one
// destination row of D = H*C channels, each lane holds VEC = D/64 contiguous channels (a 16-byte load
// for bf16 at D=512), a head spans LPH = C/VEC adjacent lanes and the per-head <q,k> dot is finished
// with a DPP butterfly inside the row.  Online softmax state (m, l) and the accumulator stay in fp32
// registers; the edge loop is software-pipelined one edge ahead.
//
// EDGE_FUSED variant: E = edge_attr @ W_e^T + b_e is never formed.  Using linearity,
//   <q_h, k_h + W_h a + b_h>       = <q_h,k_h> + sum_f a_f <q_h, W_h[:,f]> + <q_h,b_h>
//   sum_e p_e (v_h + W_h a_e + b_h) = sum_e p_e v_h + W_h (sum_e p_e a_e) + b_h sum_e p_e
// so per destination we build qw[h][f] once (W' = [W_e | b_e] staged in LDS), per edge we only touch the
// fe_pad fp32 edge features (wave-uniform -> scalar loads) and at the end apply W' to the weighted feature
// sum.  HBM traffic per layer drops from 2(4ND + MD) to 2*4ND + 4*M*fe_pad bytes.
#include <type_traits>

#include "common.h"

namespace anemoi {

constexpr int kWavesPerBlock = 4;

template <int VEC>
struct EdgeRow {
  float k[VEC];
  float v[VEC];
  float e[VEC];
};

// ---------------------------------------------------------------------------------------------- fast path
template <typename T, int VEC, int LPH, bool HAS_E>
__global__ __launch_bounds__(64 * kWavesPerBlock) void gt_attn_fwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const T* __restrict__ e, int64_t lde, const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
    const T* __restrict__ addend, int64_t ldadd, T* __restrict__ out, int64_t ldo, float* __restrict__ lse, int n_dst,
    int H, float scale) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
  const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
  const int c0 = lane * VEC;

  float qv[VEC], acc[VEC];
  load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    qv[j] *= scale;
    acc[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;

  EdgeRow<VEC> cur, nxt;
  auto fetch = [&](int ei, EdgeRow<VEC>& r) {
    const int s = __builtin_amdgcn_readfirstlane(row[ei]);
    load_vec<T, VEC>(k + (int64_t)s * ldk + c0, r.k);
    load_vec<T, VEC>(v + (int64_t)s * ldv + c0, r.v);
    if constexpr (HAS_E) load_vec<T, VEC>(e + (int64_t)ei * lde + c0, r.e);
  };
  if (beg < end) fetch(beg, nxt);
  for (int ei = beg; ei < end; ++ei) {
    cur = nxt;
    if (ei + 1 < end) fetch(ei + 1, nxt);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if constexpr (HAS_E) {
        cur.k[j] += cur.e[j];
        cur.v[j] += cur.e[j];
      }
      dot = fmaf(qv[j], cur.k[j], dot);
    }
    dot = group_sum<LPH>(dot);
    const float m_new = fmaxf(m, dot);
    const float corr = __expf(m - m_new);  // first edge: exp(-inf) = 0
    const float p = __expf(dot - m_new);
    l = fmaf(l, corr, p);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(acc[j], corr, p * cur.v[j]);
    m = m_new;
  }

  const float inv = (end > beg) ? 1.0f / l : 0.f;
  float o[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = acc[j] * inv;
  if (addend != nullptr) {
    float a[VEC];
    load_vec<T, VEC>(addend + (int64_t)d * ldadd + c0, a);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] += a[j];
  }
  store_vec<T, VEC>(out + (int64_t)d * ldo + c0, o);
  if (lse != nullptr && (lane % LPH) == 0) lse[(int64_t)d * H + lane / LPH] = (end > beg) ? m + __logf(l) : 0.f;
}

// ---------------------------------------------------------------------------------------------- fused lin_edge
// LDS image of W' = [W_e | b_e | 0]: per lane a chunk of VEC*FE_PAD floats (+4 floats of padding so that
// 16 consecutive lanes' ds_read_b128 hit 16 distinct 16-byte bank slots).
template <int VEC, int FE_PAD>
struct WLayout {
  static constexpr int kChunk = VEC * FE_PAD + 4;
  static constexpr int kFloats = 64 * kChunk;
};

template <typename T, int VEC, int LPH, int FE_PAD>
__global__ __launch_bounds__(64 * kWavesPerBlock) void gt_attn_fused_edge_fwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const float* __restrict__ feat, int fe, const T* __restrict__ w_edge, const T* __restrict__ b_edge,
    const int32_t* __restrict__ row, const int32_t* __restrict__ colptr, const T* __restrict__ addend, int64_t ldadd,
    T* __restrict__ out, int64_t ldo, float* __restrict__ lse, int n_dst, int H, float scale, int dst_per_wave) {
  using L = WLayout<VEC, FE_PAD>;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // [64][kChunk]
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VEC;

  // Stage W' once per block: element (channel c, feature f) -> w_lds[(c / VEC) * kChunk + (c % VEC) * FE_PAD + f].
  for (int i = threadIdx.x; i < 64 * VEC * FE_PAD; i += blockDim.x) {
    const int c = i / FE_PAD, f = i % FE_PAD;
    float val = 0.f;
    if (f < fe) val = to_float(w_edge[(int64_t)c * fe + f]);
    else if (f == fe && b_edge != nullptr) val = to_float(b_edge[c]);
    w_lds[(c / VEC) * L::kChunk + (c % VEC) * FE_PAD + f] = val;
  }
  __syncthreads();
  const float* wl = w_lds + lane * L::kChunk;

  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  const int d_beg = wave * dst_per_wave;
  const int d_end = min(n_dst, d_beg + dst_per_wave);

  for (int d = d_beg; d < d_end; ++d) {
    const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
    const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);

    float qv[VEC], acc[VEC];
    load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      qv[j] *= scale;
      acc[j] = 0.f;
    }
    // qw[f] = (1/LPH) * sum over the head's channels of q[c] * W'[c][f]  (pre-divided: every lane of the
    // head adds the same edge-feature term before the head butterfly).
    float qw[FE_PAD], sf[FE_PAD];
#pragma unroll
    for (int f = 0; f < FE_PAD; ++f) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) t = fmaf(qv[j], wl[j * FE_PAD + f], t);
      qw[f] = group_sum<LPH>(t) * (1.0f / LPH);
      sf[f] = 0.f;
    }
    float m = -INFINITY, l = 0.f;

    struct Row {
      float k[VEC];
      float v[VEC];
    } cur, nxt;
    auto fetch = [&](int ei, Row& r) {
      const int s = __builtin_amdgcn_readfirstlane(row[ei]);
      load_vec<T, VEC>(k + (int64_t)s * ldk + c0, r.k);
      load_vec<T, VEC>(v + (int64_t)s * ldv + c0, r.v);
    };
    if (beg < end) fetch(beg, nxt);
    for (int ei = beg; ei < end; ++ei) {
      cur = nxt;
      if (ei + 1 < end) fetch(ei + 1, nxt);
      const float* a = feat + (int64_t)ei * FE_PAD;  // wave-uniform address -> scalar loads
      float af[FE_PAD];
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) af[f] = a[f];
      float dot = 0.f;
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) dot = fmaf(af[f], qw[f], dot);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dot = fmaf(qv[j], cur.k[j], dot);
      dot = group_sum<LPH>(dot);
      const float m_new = fmaxf(m, dot);
      const float corr = __expf(m - m_new);
      const float p = __expf(dot - m_new);
      l = fmaf(l, corr, p);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(acc[j], corr, p * cur.v[j]);
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) sf[f] = fmaf(sf[f], corr, p * af[f]);
      m = m_new;
    }

    const float inv = (end > beg) ? 1.0f / l : 0.f;
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = acc[j];
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) t = fmaf(sf[f], wl[j * FE_PAD + f], t);
      o[j] = t * inv;
    }
    if (addend != nullptr) {
      float ad[VEC];
      load_vec<T, VEC>(addend + (int64_t)d * ldadd + c0, ad);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += ad[j];
    }
    store_vec<T, VEC>(out + (int64_t)d * ldo + c0, o);
    if (lse != nullptr && (lane % LPH) == 0) lse[(int64_t)d * H + lane / LPH] = (end > beg) ? m + __logf(l) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- generic path
// Any (H, C): one thread per (destination, head), serial over edges and channels.  Used for shapes the
// wave-per-row layout cannot express (D not a multiple of 64, non power-of-two lanes per head — e.g. the
// reference's own test shapes H in {2,6}, C in {4,6}).  Correctness path, not a performance path.
constexpr int kGenericMaxC = 256;

template <typename T>
__global__ void gt_attn_fwd_generic_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                           const T* __restrict__ v, int64_t ldv, const T* __restrict__ e, int64_t lde,
                                           const float* __restrict__ feat, int fe, int fe_pad,
                                           const T* __restrict__ w_edge, const T* __restrict__ b_edge,
                                           const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
                                           const T* __restrict__ addend, int64_t ldadd, T* __restrict__ out,
                                           int64_t ldo, float* __restrict__ lse, int n_dst, int H, int C, float scale) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_dst * H) return;
  const int d = (int)(t / H), h = (int)(t % H);
  const int beg = colptr[d], end = colptr[d + 1];
  float acc[kGenericMaxC];  // private (scratch-memory) fp32 accumulator
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  float m = -INFINITY, l = 0.f;
  const T* qp = q + (int64_t)d * ldq + h * C;
  for (int ei = beg; ei < end; ++ei) {
    const int s = row[ei];
    const T* kp = k + (int64_t)s * ldk + h * C;
    const T* vp = v + (int64_t)s * ldv + h * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      float ee = 0.f;
      if (e != nullptr) ee = to_float(e[(int64_t)ei * lde + h * C + c]);
      if (feat != nullptr) {
        ee = b_edge ? to_float(b_edge[h * C + c]) : 0.f;
        for (int f = 0; f < fe; ++f) ee = fmaf(feat[(int64_t)ei * fe_pad + f], to_float(w_edge[(int64_t)(h * C + c) * fe + f]), ee);
      }
      dot = fmaf(to_float(qp[c]) * scale, to_float(kp[c]) + ee, dot);
    }
    const float m_new = fmaxf(m, dot);
    const float corr = expf(m - m_new), p = expf(dot - m_new);
    l = l * corr + p;
    for (int c = 0; c < C; ++c) {
      float ee = 0.f;
      if (e != nullptr) ee = to_float(e[(int64_t)ei * lde + h * C + c]);
      if (feat != nullptr) {
        ee = b_edge ? to_float(b_edge[h * C + c]) : 0.f;
        for (int f = 0; f < fe; ++f) ee = fmaf(feat[(int64_t)ei * fe_pad + f], to_float(w_edge[(int64_t)(h * C + c) * fe + f]), ee);
      }
      acc[c] = acc[c] * corr + p * (to_float(vp[c]) + ee);
    }
    m = m_new;
  }
  const float inv = (end > beg) ? 1.0f / l : 0.f;
  for (int c = 0; c < C; ++c) {
    float o = acc[c] * inv;
    if (addend != nullptr) o += to_float(addend[(int64_t)d * ldadd + h * C + c]);
    out[(int64_t)d * ldo + h * C + c] = from_float<T>(o);
  }
  if (lse != nullptr) lse[(int64_t)d * H + h] = (end > beg) ? m + logf(l) : 0.f;
}

template <typename T>
__global__ void pack_edge_features_kernel(const T* __restrict__ ea, int64_t ld, float* __restrict__ out, int M, int fe,
                                          int fe_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * fe_pad) return;
  const int m = (int)(i / fe_pad), f = (int)(i % fe_pad);
  out[i] = f < fe ? to_float(ea[(int64_t)m * ld + f]) : (f == fe ? 1.0f : 0.0f);
}

// ---------------------------------------------------------------------------------------------- dispatch
struct AttnArgs {
  const void *q, *k, *v, *e;
  int64_t ldq, ldk, ldv, lde;
  const float* feat;
  int fe, fe_pad;
  const void *w_edge, *b_edge;
  const int32_t *row, *colptr;
  const void* addend;
  int64_t ldadd;
  void* out;
  int64_t ldo;
  float* lse;
  int n_dst, n_src, H, C;
  hipStream_t stream;
};

static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

template <typename T, int VEC, int LPH>
static int launch_fast(const AttnArgs& a) {
  const float scale = 1.0f / sqrtf((float)a.C);
  const dim3 block(64 * kWavesPerBlock);
  if (a.feat == nullptr) {
    const dim3 grid((a.n_dst + kWavesPerBlock - 1) / kWavesPerBlock);
    if (a.e != nullptr)
      hipLaunchKernelGGL((gt_attn_fwd_kernel<T, VEC, LPH, true>), grid, block, 0, a.stream, (const T*)a.q, a.ldq,
                         (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, a.row, a.colptr,
                         (const T*)a.addend, a.ldadd, (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, scale);
    else
      hipLaunchKernelGGL((gt_attn_fwd_kernel<T, VEC, LPH, false>), grid, block, 0, a.stream, (const T*)a.q, a.ldq,
                         (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)nullptr, (int64_t)0, a.row, a.colptr,
                         (const T*)a.addend, a.ldadd, (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, scale);
    return check_launch("gt_attn_fwd_kernel");
  }
  // Fused lin_edge: each wave walks `dst_per_wave` consecutive destinations so the W' staging is amortised,
  // while keeping >= ~8 blocks per CU worth of waves in flight.
  auto go = [&](auto fe_pad_c) {
    constexpr int FE_PAD = decltype(fe_pad_c)::value;
    using L = WLayout<VEC, FE_PAD>;
    if (L::kFloats * sizeof(float) > 64 * 1024) return 1;  // beyond the default dynamic-LDS limit: generic path
    int dst_per_wave = a.n_dst >= 32768 ? 8 : (a.n_dst >= 4096 ? 4 : (a.n_dst >= 1024 ? 2 : 1));
    const int waves = (a.n_dst + dst_per_wave - 1) / dst_per_wave;
    const dim3 grid((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((gt_attn_fused_edge_fwd_kernel<T, VEC, LPH, FE_PAD>), grid, block, L::kFloats * sizeof(float),
                       a.stream, (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, a.feat, a.fe,
                       (const T*)a.w_edge, (const T*)a.b_edge, a.row, a.colptr, (const T*)a.addend, a.ldadd, (T*)a.out,
                       a.ldo, a.lse, a.n_dst, a.H, scale, dst_per_wave);
    return check_launch("gt_attn_fused_edge_fwd_kernel");
  };
  switch (a.fe_pad) {
    case 4: return go(std::integral_constant<int, 4>{});
    case 8: return go(std::integral_constant<int, 8>{});
    case 12: return go(std::integral_constant<int, 12>{});
    case 16: return go(std::integral_constant<int, 16>{});
    default: return 1;  // not covered by the fast path
  }
}

template <typename T, int VEC>
static int launch_vec(const AttnArgs& a) {
  const int lph = a.C / VEC;
  switch (lph) {
    case 1: return launch_fast<T, VEC, 1>(a);
    case 2: return launch_fast<T, VEC, 2>(a);
    case 4: return launch_fast<T, VEC, 4>(a);
    case 8: return launch_fast<T, VEC, 8>(a);
    case 16: return launch_fast<T, VEC, 16>(a);
    default: return 1;
  }
}

template <typename T>
static int launch(const AttnArgs& a) {
  const int D = a.H * a.C;
  int rc = 1;
  // fast path: one wave covers the row exactly, rows 16-byte aligned for vector access
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool lds_ok = (a.ldq % 8 == 0) && (a.ldk % 8 == 0) && (a.ldv % 8 == 0) && (a.ldo % 8 == 0) &&
                      (a.e == nullptr || a.lde % 8 == 0) && (a.addend == nullptr || a.ldadd % 8 == 0) && al16(a.q) &&
                      al16(a.k) && al16(a.v) && al16(a.out) && al16(a.e) && al16(a.addend) && al16(a.feat);
  if (D % 64 == 0 && lds_ok) {
    const int vec = D / 64;
    if (a.C % vec == 0 && pow2(a.C / vec) && a.C / vec <= 16) {
      if (vec == 1) rc = launch_vec<T, 1>(a);
      else if (vec == 2) rc = launch_vec<T, 2>(a);
      else if (vec == 4) rc = launch_vec<T, 4>(a);
      else if (vec == 8) rc = launch_vec<T, 8>(a);
      else if (vec == 16) rc = launch_vec<T, 16>(a);
    }
  }
  if (rc <= 0) return rc;
  // generic path
  if (a.C > kGenericMaxC) {
    set_error("gt_attention: channels per head C=%d not supported (fast path needs H*C %% 64 == 0 and a power-of-two "
              "C*64/(H*C) <= 16; generic path needs C <= %d)", a.C, kGenericMaxC);
    return ANEMOI_E_UNSUPPORTED;
  }
  const int64_t threads = (int64_t)a.n_dst * a.H;
  if (threads == 0) return ANEMOI_OK;
  const float scale = 1.0f / sqrtf((float)a.C);
  hipLaunchKernelGGL((gt_attn_fwd_generic_kernel<T>), dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, a.stream,
                     (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, a.feat, a.fe,
                     a.fe_pad, (const T*)a.w_edge, (const T*)a.b_edge, a.row, a.colptr, (const T*)a.addend, a.ldadd,
                     (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, a.C, scale);
  return check_launch("gt_attn_fwd_generic_kernel");
}

static int dispatch(const AttnArgs& a, anemoi_dtype_t dtype) {
  switch (dtype) {
    case ANEMOI_F32: return launch<float>(a);
    case ANEMOI_BF16: return launch<bf16_t>(a);
    case ANEMOI_F16: return launch<f16_t>(a);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const int32_t* row, const int32_t* colptr,
                                       const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                                       int32_t n_dst, int32_t n_src, int32_t H, int32_t C, anemoi_dtype_t dtype,
                                       void* stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && H > 0 && C > 0, "gt_attention_fwd: bad sizes n_dst=%d n_src=%d H=%d C=%d", n_dst, n_src, H, C);
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(q && k && v && out && colptr, "gt_attention_fwd: null q/k/v/out/colptr");
  const int D = H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D && (!e || lde >= D) && (!addend || ldadd >= D),
                 "gt_attention_fwd: leading dimension smaller than H*C=%d", D);
  AttnArgs a{q, k, v, e, ldq, ldk, ldv, lde, nullptr, 0, 0, nullptr, nullptr, row, colptr, addend, ldadd, out, ldo, lse, n_dst, n_src, H, C, as_stream(stream)};
  return dispatch(a, dtype);
}

extern "C" int anemoi_gt_attention_fused_edge_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                                  int64_t ldv, const float* edge_feat, int32_t fe, int32_t fe_pad,
                                                  const void* w_edge, const void* b_edge, const int32_t* row,
                                                  const int32_t* colptr, const void* addend, int64_t ldadd, void* out,
                                                  int64_t ldo, float* lse, int32_t n_dst, int32_t n_src, int32_t H,
                                                  int32_t C, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && H > 0 && C > 0, "gt_attention_fused_edge_fwd: bad sizes");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(q && k && v && out && colptr && edge_feat && w_edge, "gt_attention_fused_edge_fwd: null pointer");
  ANEMOI_REQUIRE(fe > 0 && fe_pad == 4 * ((fe + 1 + 3) / 4), "gt_attention_fused_edge_fwd: fe_pad must be 4*ceil((fe+1)/4), got fe=%d fe_pad=%d", fe, fe_pad);
  const int D = H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D && (!addend || ldadd >= D), "gt_attention_fused_edge_fwd: leading dimension smaller than H*C=%d", D);
  AttnArgs a{q, k, v, nullptr, ldq, ldk, ldv, 0, edge_feat, fe, fe_pad, w_edge, b_edge, row, colptr, addend, ldadd, out, ldo, lse, n_dst, n_src, H, C, as_stream(stream)};
  return dispatch(a, dtype);
}

extern "C" int anemoi_pack_edge_features(const void* edge_attr, int64_t ld, float* out, int32_t M, int32_t fe,
                                         int32_t fe_pad, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(M >= 0 && fe > 0 && fe_pad == 4 * ((fe + 1 + 3) / 4) && ld >= fe, "pack_edge_features: bad sizes M=%d fe=%d fe_pad=%d", M, fe, fe_pad);
  if (M == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(edge_attr && out, "pack_edge_features: null pointer");
  const int64_t n = (int64_t)M * fe_pad;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((pack_edge_features_kernel<float>), grid, block, 0, as_stream(stream), (const float*)edge_attr, ld, out, M, fe, fe_pad); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((pack_edge_features_kernel<bf16_t>), grid, block, 0, as_stream(stream), (const bf16_t*)edge_attr, ld, out, M, fe, fe_pad); break;
    case ANEMOI_F16: hipLaunchKernelGGL((pack_edge_features_kernel<f16_t>), grid, block, 0, as_stream(stream), (const f16_t*)edge_attr, ld, out, M, fe, fe_pad); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("pack_edge_features_kernel");
}
