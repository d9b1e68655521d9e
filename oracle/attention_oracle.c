/*
 * attention_oracle.c — CPU restatement of the reference's attention forward.  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The shipped product (libflash_hip.so and everything above it) never does.
 *
 * Parity status: PINNED.  The functions here are checked (tests/test_oracle.py) against golden
 * vectors in tests/golden/ that are outputs of the reference's own oracle function: the generating
 * script tests/golden/make_golden.py takes the definition of `manual_attn` out of the reference's
 * benchmarks/bench.py (lines 39-43; the same expression as main.cpp:12-17) and executes that text,
 * unmodified, against the local torch on the reference's test inputs (manual_seed(0), randn q, k, v
 * in that order, main.cpp:24-34) and test shapes (main.cpp:157-193).  `make_golden.py --check`
 * (run by tests/test_oracle.py wherever the reference tree is present) regenerates the vectors and
 * requires the committed files to be bit-identical.
 *
 * Two algorithms:
 *   oracle_attn_naive      — the reference's oracle: S = Q K^T * scale, P = softmax(S), O = P V,
 *                            fp32 throughout (main.cpp:12-17).  Also the timed CPU baseline.
 *   oracle_attn_tiled      — the reference's hot kernel as a scalar loop nest: Br x Bc tiles, key
 *                            and query tails cropped (Bcc/Brc), per-tile row max / row sum, the
 *                            (m, l) recurrence and the per-tile renormalisation of O
 *                            (flash/kernel_mma_qreg.cuh:100-266, recurrence :253-263, O update
 *                            :220-233; tile sizes flash/KernelParameters.hpp:340-342).  Operand
 *                            precision is selectable so that the same loop models the TF32
 *                            reference (fp32 here) and the fp16-operand MI355X kernel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- IEEE binary16 <-> binary32, round-to-nearest-even, software (no F16C dependence) -------- */
static uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t aexp = (x >> 23) & 0xffu;
  uint32_t mant = x & 0x7fffffu;
  if (aexp == 0xffu) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0u));
  int e = (int)aexp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x800000u;
    const int shift = 14 - e; /* 14..24 */
    uint32_t hm = mant >> shift;
    const uint32_t remn = mant & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (remn > half || (remn == half && (hm & 1u))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
  const uint32_t remn = mant & 0x1fffu;
  if (remn > 0x1000u || (remn == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

static float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {
      int sh = 0;
      while (!(m & 0x400u)) {
        m <<= 1;
        sh++;
      }
      m &= 0x3ffu;
      x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e - 15 + 127) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

float oracle_round_f16(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }

void oracle_round_f16_array(const float* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = oracle_round_f16(in[i]);
}

void oracle_f32_to_f16_bits(const float* in, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = f32_to_f16_bits(in[i]);
}

void oracle_f16_bits_to_f32(const uint16_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = f16_bits_to_f32(in[i]);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/*
 * Reference oracle (main.cpp:12-17): att = q @ k^T * scale; att = softmax(att, -1); y = att @ v.
 * q, k, v, o: contiguous [B, H, N, d] fp32.  One (b, h) head per OpenMP task; per query row the
 * N scores live in a thread-private buffer.  Returns 0, or -1 if a scratch allocation failed.
 */
int oracle_attn_naive(const float* q, const float* k, const float* v, float* o, int64_t B, int64_t H,
                      int64_t N, int64_t d, float scale) {
  const int64_t heads = B * H;
  int failed = 0;
#pragma omp parallel
  {
    float* s = (float*)malloc((size_t)N * sizeof(float));
    float* acc = (float*)malloc((size_t)d * sizeof(float));
    if (!s || !acc) {
#pragma omp atomic write
      failed = 1;
    }
#pragma omp for collapse(2) schedule(dynamic, 8)
    for (int64_t bh = 0; bh < heads; ++bh) {
      for (int64_t i = 0; i < N; ++i) {
        if (!s || !acc) continue;
        const float* qi = q + (bh * N + i) * d;
        const float* kh = k + bh * N * d;
        const float* vh = v + bh * N * d;
        float mx = -INFINITY;
        for (int64_t j = 0; j < N; ++j) {
          const float* kj = kh + j * d;
          float dot = 0.f;
#pragma omp simd reduction(+ : dot)
          for (int64_t x = 0; x < d; ++x) dot += qi[x] * kj[x];
          dot *= scale;
          s[j] = dot;
          mx = dot > mx ? dot : mx;
        }
        float sum = 0.f;
        for (int64_t j = 0; j < N; ++j) {
          const float e = expf(s[j] - mx);
          s[j] = e;
          sum += e;
        }
        const float inv = 1.0f / sum;
        for (int64_t x = 0; x < d; ++x) acc[x] = 0.f;
        for (int64_t j = 0; j < N; ++j) {
          const float pj = s[j] * inv;
          const float* vj = vh + j * d;
#pragma omp simd
          for (int64_t x = 0; x < d; ++x) acc[x] += pj * vj[x];
        }
        float* oi = o + (bh * N + i) * d;
        for (int64_t x = 0; x < d; ++x) oi[x] = acc[x];
      }
    }
    free(s);
    free(acc);
  }
  return failed ? -1 : 0;
}

/*
 * The hot kernel's loop nest (flash/kernel_mma_qreg.cuh:100-266) in scalar form.
 *   Br, Bc            tile sizes (reference: 128, 64, KernelParameters.hpp:340-342)
 *   p_f16             nonzero: round the probabilities P to fp16 before the P V product (models the
 *                     fp16 MFMA operand of the MI355X kernel); zero: keep fp32 (models TF32 closely
 *                     enough for the oracle's purpose)
 *   defer_norm        zero: divide O by the running row sum after every K/V tile exactly as the
 *                     reference does (:220-233); nonzero: keep O unnormalised and divide once at
 *                     the end (what the MI355X kernel does)
 *   m_out, l_out      optional [B, H, N] final row max (of scaled scores) and row sum
 *   q_prescale_f16    (oracle_attn_tiled_q16 only) nonzero: model the one discretionary rounding of
 *                     the MI355X kernel: the query row is multiplied by scale*log2(e) and rounded to
 *                     fp16 once, scores are the plain dot products of that row, probabilities are
 *                     exp2(score - max).  m_out is then reported in the same log2 domain.
 * Inputs are used as given; callers that model fp16 operands pass fp16-rounded values.
 */
static int attn_tiled_impl(const float* q, const float* k, const float* v, float* o, int64_t B, int64_t H, int64_t N,
                           int64_t d, float scale, int Br, int Bc, int p_f16, int defer_norm, int q_prescale_f16,
                           float* m_out, float* l_out) {
  if (Br <= 0 || Bc <= 0) return -2;
  const float kLog2e = 1.44269504088896340736f;
  const int64_t heads = B * H;
  const int64_t nqt = (N + Br - 1) / Br;
  int failed = 0;
#pragma omp parallel
  {
    float* S = (float*)malloc((size_t)Br * Bc * sizeof(float));
    float* Oacc = (float*)malloc((size_t)Br * d * sizeof(float));
    float* mprev = (float*)malloc((size_t)Br * sizeof(float));
    float* lprev = (float*)malloc((size_t)Br * sizeof(float));
    float* qrow = (float*)malloc((size_t)(d > 0 ? d : 1) * sizeof(float));
    if (!S || !Oacc || !mprev || !lprev || !qrow) {
#pragma omp atomic write
      failed = 1;
    }
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int64_t bh = 0; bh < heads; ++bh) {
      for (int64_t it = 0; it < nqt; ++it) {
        if (!S || !Oacc || !mprev || !lprev || !qrow) continue;
        const float* qh = q + bh * N * d;
        const float* kh = k + bh * N * d;
        const float* vh = v + bh * N * d;
        const int64_t iStart = it * Br;
        const int Brc = (int)((N - iStart) < Br ? (N - iStart) : Br); /* cropped query tile (:57) */
        for (int ii = 0; ii < Brc; ++ii) {
          mprev[ii] = -INFINITY; /* :70-73 */
          lprev[ii] = 0.f;
          for (int64_t x = 0; x < d; ++x) Oacc[ii * d + x] = 0.f; /* O = zeros_like(Q), launch.cu:195 */
        }
        for (int64_t jStart = 0; jStart < N; jStart += Bc) {            /* :100 */
          const int Bcc = (int)((N - jStart) < Bc ? (N - jStart) : Bc); /* :102 */
          for (int ii = 0; ii < Brc; ++ii) {
            const float* qi = qh + (iStart + ii) * d;
            if (q_prescale_f16) {
              for (int64_t x = 0; x < d; ++x) qrow[x] = oracle_round_f16(qi[x] * (scale * kLog2e));
              qi = qrow;
            }
            /* S = Q K^T * scale, row max (:148-168) */
            float mcur = -INFINITY;
            for (int jj = 0; jj < Bcc; ++jj) {
              const float* kj = kh + (jStart + jj) * d;
              float dot = 0.f;
              for (int64_t x = 0; x < d; ++x) dot += qi[x] * kj[x];
              if (!q_prescale_f16) dot *= scale;
              S[ii * Bc + jj] = dot;
              mcur = dot > mcur ? dot : mcur;
            }
            /* P = exp(S - m), row sum (:173-188) */
            float lcur = 0.f;
            for (int jj = 0; jj < Bcc; ++jj) {
              float pij = q_prescale_f16 ? exp2f(S[ii * Bc + jj] - mcur) : expf(S[ii * Bc + jj] - mcur);
              lcur += pij;
              if (p_f16) pij = oracle_round_f16(pij);
              S[ii * Bc + jj] = pij;
            }
            /* (m, l) recurrence (:253-263) */
            const float mnew = mprev[ii] > mcur ? mprev[ii] : mcur;
            const float a_prev = q_prescale_f16 ? exp2f(mprev[ii] - mnew) : expf(mprev[ii] - mnew);
            const float a_cur = q_prescale_f16 ? exp2f(mcur - mnew) : expf(mcur - mnew);
            const float lnew = a_prev * lprev[ii] + a_cur * lcur;
            /* O update (:194-233): PV for this tile, then rescale */
            for (int64_t x = 0; x < d; ++x) {
              float pv = 0.f;
              for (int jj = 0; jj < Bcc; ++jj) pv += S[ii * Bc + jj] * vh[(jStart + jj) * d + x];
              float* ox = &Oacc[ii * d + x];
              if (defer_norm)
                *ox = a_prev * (*ox) + a_cur * pv;
              else
                *ox = (lprev[ii] * a_prev * (*ox) + a_cur * pv) / lnew;
            }
            mprev[ii] = mnew;
            lprev[ii] = lnew;
          }
        }
        for (int ii = 0; ii < Brc; ++ii) {
          float* oi = o + (bh * N + iStart + ii) * d;
          const float inv = defer_norm ? 1.0f / lprev[ii] : 1.0f;
          for (int64_t x = 0; x < d; ++x) oi[x] = Oacc[ii * d + x] * inv;
          if (m_out) m_out[bh * N + iStart + ii] = mprev[ii];
          if (l_out) l_out[bh * N + iStart + ii] = lprev[ii];
        }
      }
    }
    free(S);
    free(Oacc);
    free(mprev);
    free(lprev);
    free(qrow);
  }
  return failed ? -1 : 0;
}

int oracle_attn_tiled(const float* q, const float* k, const float* v, float* o, int64_t B, int64_t H, int64_t N,
                      int64_t d, float scale, int Br, int Bc, int p_f16, int defer_norm, float* m_out,
                      float* l_out) {
  return attn_tiled_impl(q, k, v, o, B, H, N, d, scale, Br, Bc, p_f16, defer_norm, 0, m_out, l_out);
}

int oracle_attn_tiled_q16(const float* q, const float* k, const float* v, float* o, int64_t B, int64_t H, int64_t N,
                          int64_t d, float scale, int Br, int Bc, int p_f16, int defer_norm, int q_prescale_f16,
                          float* m_out, float* l_out) {
  return attn_tiled_impl(q, k, v, o, B, H, N, d, scale, Br, Bc, p_f16, defer_norm, q_prescale_f16, m_out, l_out);
}
