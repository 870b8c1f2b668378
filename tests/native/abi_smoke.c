/* Native smoke test of the C ABI (no Python, no torch): links liblrx.so, drives lrx_flat_ip_search, lrx_flat_ip_search_bounded
 * and lrx_embedding_bag_mean with plain HIP allocations and checks them against brute force on the host; the varlen causal attention
 * without and with a work list (ABI 7) against a double-precision softmax on the host.
 * build: hipcc -x hip --offload-arch=gfx950 tests/native/abi_smoke.c -Iinclude -Llightretriever_amd -llrx -Wl,-rpath,$PWD/lightretriever_amd -o abi_smoke
 * (compiled and run by tests/test_gpu_native.py) */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lrx.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define LRX(x) do { int r_ = (x); if (r_ != 0) { printf("lrx error %d: %s (%s:%d)\n", r_, lrx_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static float frand(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (float)((rng_state >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f; }

/* fp32 -> fp16 bits, round to nearest even, saturating at +-65504 (what the shadow holds; include/lrx.h) */
static unsigned short f32_to_f16(float f) {
  if (f > 65504.f) f = 65504.f;
  if (f < -65504.f) f = -65504.f;
  unsigned u; memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  const int e = (int)((u >> 23) & 0xFF) - 127 + 15;
  unsigned m = u & 0x7FFFFFu;
  if (e >= 31) return (unsigned short)(sign | 0x7BFFu);
  if (e <= 0) {                                  /* subnormal half (or zero) */
    if (e < -10) return (unsigned short)sign;
    m |= 0x800000u;
    const int sh = 14 - e;                       /* 24-bit significand -> 10 bits + exponent offset */
    unsigned h = m >> sh;
    const unsigned rem = m & ((1u << sh) - 1u), halfway = 1u << (sh - 1);
    if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
    return (unsigned short)(sign | h);
  }
  unsigned h = ((unsigned)e << 10) | (m >> 13);
  const unsigned rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;      /* carries into the exponent correctly; cannot pass 0x7BFF after the clamp */
  return (unsigned short)(sign | h);
}

static int check_topk(const float* X, const float* q, int N, int D, int Q, int k, const float* Dg, const long long* Ig, const char* what) {
  double* sc = (double*)malloc(sizeof(double) * N);
  int bad = 0;
  for (int qi = 0; qi < Q; ++qi) {
    for (int n = 0; n < N; ++n) { double s = 0; for (int d = 0; d < D; ++d) s += (double)q[qi * D + d] * (double)X[(size_t)n * D + d]; sc[n] = s; }
    char* used = (char*)calloc(N, 1);
    for (int j = 0; j < k; ++j) {           /* exact top-k in double, ties -> lower row */
      int best = -1;
      for (int n = 0; n < N; ++n) if (!used[n] && (best < 0 || sc[n] > sc[best])) best = n;
      used[best] = 1;
      const long long got = Ig[qi * k + j];
      if (got != best && fabs(sc[got] - sc[best]) > 1e-6) { if (bad < 5) printf("%s: q%d rank %d: got row %lld (%.8f) want %d (%.8f)\n", what, qi, j, got, sc[got], best, sc[best]); ++bad; }
      if (fabs((double)Dg[qi * k + j] - sc[got]) > 2e-6) { if (bad < 5) printf("%s: q%d rank %d: score %.8f vs %.8f\n", what, qi, j, Dg[qi * k + j], sc[got]); ++bad; }
    }
    free(used);
  }
  free(sc);
  return bad;
}

static float f16_to_f32(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
  unsigned u;
  if (e == 0) { float f = (float)m * (1.0f / 16777216.0f); memcpy(&u, &f, 4); u |= sign; }    /* subnormal: m * 2^-24 */
  else u = sign | ((e - 15 + 127) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static float bf16_to_f32(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

/* Varlen causal GQA attention: lrx_attn_varlen_causal (no scratch) and the same launch on a work list built once (ABI 7) must agree bit for
 * bit, and both with softmax(q k^T / sqrt(d)) v computed in double from the fp16 inputs (output bf16: 2^-8 relative). */
static int attention_smoke(void) {
  const int nq = 8, nkv = 2, d = 128, lens[4] = {1, 70, 64, 131};
  int cu[5] = {0, 0, 0, 0, 0}, T, maxlen = 0;
  for (int i = 0; i < 4; ++i) { cu[i + 1] = cu[i] + lens[i]; if (lens[i] > maxlen) maxlen = lens[i]; }
  T = cu[4];
  const int W = (nq + 2 * nkv) * d;
  unsigned short* qkv = (unsigned short*)malloc(2 * (size_t)T * W);
  for (size_t i = 0; i < (size_t)T * W; ++i) qkv[i] = f32_to_f16(frand());
  void *dqkv, *dout1, *dout2, *ditems; int* dcu;
  CHECK(hipMalloc(&dqkv, 2 * (size_t)T * W)); CHECK(hipMalloc(&dout1, 2 * (size_t)T * nq * d)); CHECK(hipMalloc(&dout2, 2 * (size_t)T * nq * d));
  CHECK(hipMalloc((void**)&dcu, sizeof(cu)));
  CHECK(hipMemcpy(dqkv, qkv, 2 * (size_t)T * W, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dcu, cu, sizeof(cu), hipMemcpyHostToDevice));
  const size_t ib = lrx_attn_items_bytes(4, T, maxlen, nq, nkv, d, 0);
  if (ib == 0) { printf("lrx_attn_items_bytes returned 0\n"); return 1; }
  CHECK(hipMalloc(&ditems, ib));
  LRX(lrx_attn_varlen_causal(dqkv, dcu, 4, T, maxlen, nq, nkv, d, dout1, 0, NULL));
  LRX(lrx_attn_build_items(dcu, 4, T, maxlen, nq, nkv, d, 0, ditems, ib, NULL));
  LRX(lrx_attn_varlen_causal_items(dqkv, dcu, ditems, ib, 4, T, maxlen, nq, nkv, d, dout2, 0, NULL));
  CHECK(hipDeviceSynchronize());
  unsigned short* o1 = (unsigned short*)malloc(2 * (size_t)T * nq * d);
  unsigned short* o2 = (unsigned short*)malloc(2 * (size_t)T * nq * d);
  CHECK(hipMemcpy(o1, dout1, 2 * (size_t)T * nq * d, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2, dout2, 2 * (size_t)T * nq * d, hipMemcpyDeviceToHost));
  int bad = 0, ovf = -1;
  if (memcmp(o1, o2, 2 * (size_t)T * nq * d) != 0) { printf("attention: the work-list launch differs from the launch without a list\n"); ++bad; }
  LRX(lrx_debug_attn_items_overflow(&ovf));
  if (ovf != 0) { printf("attention: work-list builder overflow count %d\n", ovf); ++bad; }
  double* sc = (double*)malloc(sizeof(double) * maxlen);
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < lens[b]; ++i)
      for (int h = 0; h < nq; ++h) {
        const unsigned short* qr = qkv + (size_t)(cu[b] + i) * W + h * d;
        const int hk = h / (nq / nkv);
        double mx = -1e300, den = 0;
        for (int j = 0; j <= i; ++j) {
          const unsigned short* kr = qkv + (size_t)(cu[b] + j) * W + (nq + hk) * d;
          double s = 0;
          for (int e = 0; e < d; ++e) s += (double)f16_to_f32(qr[e]) * (double)f16_to_f32(kr[e]);
          sc[j] = s / sqrt((double)d);
          if (sc[j] > mx) mx = sc[j];
        }
        for (int j = 0; j <= i; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
        for (int e = 0; e < d; e += 37) {            /* a few columns per row */
          double acc = 0;
          for (int j = 0; j <= i; ++j) acc += sc[j] * (double)f16_to_f32(qkv[(size_t)(cu[b] + j) * W + (nq + nkv + hk) * d + e]);
          acc /= den;
          const double got = bf16_to_f32(o2[(size_t)(cu[b] + i) * nq * d + h * d + e]);
          if (fabs(got - acc) > 0.01 + 0.01 * fabs(acc)) { if (bad < 5) printf("attention: seq %d row %d head %d col %d: %.6f vs %.6f\n", b, i, h, e, got, acc); ++bad; }
        }
      }
  free(sc); free(o1); free(o2); free(qkv);
  return bad;
}

int main(void) {
  if (lrx_abi_version() != LRX_ABI_VERSION) { printf("ABI version mismatch: library %d, header %d\n", lrx_abi_version(), LRX_ABI_VERSION); return 1; }
  const int N = 6000, D = 64, Q = 40, k = 10;
  float* X = (float*)malloc(sizeof(float) * N * D);
  float* q = (float*)malloc(sizeof(float) * Q * D);
  float maxn = 0.f;
  for (int n = 0; n < N; ++n) { double s = 0; for (int d = 0; d < D; ++d) { X[n * D + d] = frand(); s += X[n * D + d] * X[n * D + d]; } if (sqrt(s) > maxn) maxn = (float)sqrt(s); }
  for (int i = 0; i < Q * D; ++i) q[i] = frand();
  /* the tiled fp16 shadow (round to nearest even; layout formula of include/lrx.h) made on the host */
  const size_t NSH = ((size_t)(N + 127) / 128) * 128 * D;
  unsigned short* Xb = (unsigned short*)calloc(NSH, 2);
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < D; ++c)
      Xb[((size_t)(r / 128) * (D / 64) + c / 64) * 8192 + ((((r / 16) % 8) * 2 + (c / 32) % 2) * 64 + ((c / 8) % 4) * 16 + r % 16) * 8 + c % 8] = f32_to_f16(X[r * D + c]);
  float *dX, *dq, *dD, *dbound; long long* dI; void *dXb, *ws;
  maxn *= 1.000001f;
  CHECK(hipMalloc((void**)&dX, sizeof(float) * N * D)); CHECK(hipMalloc((void**)&dq, sizeof(float) * Q * D));
  CHECK(hipMalloc((void**)&dD, sizeof(float) * Q * k)); CHECK(hipMalloc((void**)&dI, sizeof(long long) * Q * k));
  CHECK(hipMalloc(&dXb, 2 * NSH)); CHECK(hipMemset(dXb, 0, 2 * NSH)); CHECK(hipMalloc((void**)&dbound, 8)); CHECK(hipMemset(dbound, 0, 8));
  CHECK(hipMemcpy(dX, X, sizeof(float) * N * D, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dq, q, sizeof(float) * Q * D, hipMemcpyHostToDevice));
  /* shard maintenance on the device: fp16 shadow + the bounds {max |row|, max |row - fp16(row)|}; checked against the host copies */
  LRX(lrx_shard_commit_rows(dX, D, N, D, dXb, 0, dbound, NULL));
  CHECK(hipDeviceSynchronize());
  { unsigned short* Xb2 = (unsigned short*)malloc(2 * NSH); float hb[2];
    CHECK(hipMemcpy(Xb2, dXb, 2 * NSH, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb, dbound, 8, hipMemcpyDeviceToHost));
    if (memcmp(Xb, Xb2, 2 * NSH) != 0) { printf("lrx_shard_commit_rows: shadow differs from host RNE / the documented tile layout\n"); return 5; }
    if (fabsf(hb[0] / maxn - 1.f) > 1e-5f || !(hb[1] > 0.f && hb[1] < hb[0] / 1500.f)) { printf("lrx_shard_commit_rows: bounds %g %g (host max norm %g)\n", hb[0], hb[1], maxn); return 5; }
    free(Xb2); }
  const size_t wsb = lrx_flat_ip_bounded_workspace_bytes(N, D, Q, k, LRX_SEARCH_FILTER_AUTO);
  CHECK(hipMalloc(&ws, wsb)); CHECK(hipMemset(ws, 0, wsb));
  float* hD = (float*)malloc(sizeof(float) * Q * k); long long* hI = (long long*)malloc(sizeof(long long) * Q * k);
  int bad = 0;
  LRX(lrx_flat_ip_search(dX, N, D, D, NULL, dq, Q, k, 0, dD, (int64_t*)dI, ws, wsb, NULL));      /* bounds not handed over: measured by the call */
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hD, dD, sizeof(float) * Q * k, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hI, dI, sizeof(long long) * Q * k, hipMemcpyDeviceToHost));
  bad += check_topk(X, q, N, D, Q, k, hD, hI, "lrx_flat_ip_search");
  LRX(lrx_flat_ip_search_bounded(dX, N, D, D, dXb, dbound, dq, Q, k, 0, dD, (int64_t*)dI, ws, wsb, LRX_SEARCH_FILTER_AUTO, NULL));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hD, dD, sizeof(float) * Q * k, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hI, dI, sizeof(long long) * Q * k, hipMemcpyDeviceToHost));
  bad += check_topk(X, q, N, D, Q, k, hD, hI, "lrx_flat_ip_search_bounded");
  /* argument errors come back as codes + message, never as a crash */
  if (lrx_flat_ip_search(dX, N, D, D, dbound, dq, Q, 0, 0, dD, (int64_t*)dI, ws, wsb, NULL) == 0 || strlen(lrx_last_error()) == 0) { printf("k = 0 was accepted\n"); ++bad; }
  if (lrx_flat_ip_search(dX, N, D, D, dbound, dq, Q, k, 0, dD, (int64_t*)dI, ws, 16, NULL) == 0) { printf("a 16-byte workspace was accepted\n"); ++bad; }
  if (lrx_flat_ip_search_bounded(dX, N, D, D, dXb, dbound, dq, Q, k, 0, dD, (int64_t*)dI, ws, wsb, 64, NULL) == 0) { printf("unknown search flags were accepted\n"); ++bad; }
  /* (ABI 8) how many queries one pass over the shadow serves: 256 for small / narrow shards, wide chunks over a wide shadow */
  if (lrx_flat_ip_bounded_chunk_queries(N, D, 1000, k, LRX_SEARCH_FILTER_AUTO, 1) != 256 || lrx_flat_ip_bounded_chunk_queries(1000000, 2048, 1000, 100, LRX_SEARCH_FILTER_AUTO, 1) != 1008 ||
      lrx_flat_ip_bounded_chunk_queries(1000000, 2048, 1000, 100, LRX_SEARCH_FILTER_AUTO, 0) != 128) { printf("lrx_flat_ip_bounded_chunk_queries: unexpected chunking\n"); ++bad; }
  /* (ABI 8) pooling strategies on rows that need no model: mean of two unit-RMS rows through an all-ones norm weight, no normalisation */
  { const int H = 64; float hx[3 * 64]; unsigned short w16[64]; int hcu[3] = {0, 2, 3}; float ho[2 * 64];
    for (int i = 0; i < H; ++i) { hx[i] = (i & 1) ? 1.f : -1.f; hx[H + i] = 1.f; hx[2 * H + i] = (i & 2) ? 1.f : -1.f; w16[i] = 0x3F80; }   /* bf16 1.0 */
    float* dx; void* dw; int* dc; float* dout;
    CHECK(hipMalloc((void**)&dx, sizeof(hx))); CHECK(hipMalloc(&dw, sizeof(w16))); CHECK(hipMalloc((void**)&dc, sizeof(hcu))); CHECK(hipMalloc((void**)&dout, sizeof(ho)));
    CHECK(hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, w16, sizeof(w16), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, hcu, sizeof(hcu), hipMemcpyHostToDevice));
    LRX(lrx_pool_norm_mode(dx, dw, dc, 2, H, 0.f, LRX_POOL_MEAN, dout, H, H, 0, NULL, 0, NULL, 1, NULL));
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    for (int i = 0; i < H; ++i) {
      const float want0 = ((i & 1) ? 1.f : 0.f), want1 = (i & 2) ? 1.f : -1.f;                 /* mean(-1|+1, +1) and the single row of sequence 1 */
      if (fabsf(ho[i] - want0) > 1e-5f || fabsf(ho[H + i] - want1) > 1e-5f) { if (bad < 5) printf("lrx_pool_norm_mode(mean): column %d: %g %g\n", i, ho[i], ho[H + i]); ++bad; }
    }
    if (lrx_pool_norm_mode(dx, dw, dc, 2, H, 0.f, 9, dout, H, H, 0, NULL, 0, NULL, 1, NULL) == 0) { printf("an unknown pooling strategy was accepted\n"); ++bad; }
    hipFree(dx); hipFree(dw); hipFree(dc); hipFree(dout); }
  bad += attention_smoke();
  printf(bad ? "ABI SMOKE FAILED (%d mismatches)\n" : "ABI SMOKE OK\n", bad);
  return bad ? 4 : 0;
}
