// tools/microbench2.hip -- second round of arithmetic probes: candidate butterflies / modmuls.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;
constexpr int ITERS = 1024;
constexpr int ILP = 8;

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

// A: baseline Harvey/Shoup butterfly, twiddle varies per iteration
__global__ void k_bfly_int(u64* out, const u64* tw, u64 q) {
  u64 v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (threadIdx.x * 977 + i * 13) % q;
  const u64 q2 = q << 1;
  for (int it = 0; it < ITERS; it++) {
    const u64 w = tw[2 * (it & 63)], wq = tw[2 * (it & 63) + 1];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      u64 &X = v[2 * i], &Y = v[2 * i + 1];
      u64 x = X >= q2 ? X - q2 : X;
      u64 t = Y * w - mulhi64(Y, wq) * q;
      X = x + t; Y = x + q2 - t;
    }
  }
  u64 s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// B: FP64 signed butterfly (q < 2^50): T = Y*W - rint(Y*W/q)*q exactly via fma error-free product
__global__ void k_bfly_f64(double* out, const double* tw, double q) {
  double v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (double)((threadIdx.x * 977 + i * 13) % 100000);
  for (int it = 0; it < ITERS; it++) {
    const double w = tw[2 * (it & 63)], wq = tw[2 * (it & 63) + 1];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      double &X = v[2 * i], &Y = v[2 * i + 1];
      double qf = rint(Y * wq);
      double xh = Y * w;
      double xl = fma(Y, w, -xh);
      double t = fma(-qf, q, xh) + xl;
      double x = X;
      X = x + t; Y = x - t;
    }
    if ((it & 3) == 3) {  // periodic range reduction as the real kernel would do every few stages
#pragma unroll
      for (int i = 0; i < 2 * ILP; i++) v[i] = fma(-rint(v[i] * (1.0 / q)), q, v[i]);
    }
  }
  double s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// B2: FP64 butterfly without periodic reduction
__global__ void k_bfly_f64_nored(double* out, const double* tw, double q) {
  double v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (double)((threadIdx.x * 977 + i * 13) % 100000);
  for (int it = 0; it < ITERS; it++) {
    const double w = tw[2 * (it & 63)], wq = tw[2 * (it & 63) + 1];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      double &X = v[2 * i], &Y = v[2 * i + 1];
      double qf = rint(Y * wq);
      double xh = Y * w;
      double xl = fma(Y, w, -xh);
      double t = fma(-qf, q, xh) + xl;
      double x = X;
      X = x + t; Y = (x - t) * 0.5;  // keep bounded without changing op count much
    }
  }
  double s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// C: variable x variable modmul, two-word Barrett (current reduce128)
__device__ __forceinline__ u64 reduce128(u128 x, u64 q, u64 r0, u64 r1) {
  const u64 x0 = (u64)x, x1 = (u64)(x >> 64);
  u64 carry = mulhi64(x0, r0);
  u128 t2 = (u128)x0 * r1;
  u64 t1 = (u64)t2 + carry;
  u64 t3 = (u64)(t2 >> 64) + (t1 < (u64)t2);
  u128 t4 = (u128)x1 * r0;
  u64 t5 = t1 + (u64)t4;
  carry = (u64)(t4 >> 64) + (t5 < t1);
  u64 qhat = x1 * r1 + t3 + carry;
  u64 r = x0 - qhat * q;
  return r >= q ? r - q : r;
}
__global__ void k_mul_barrett2(u64* out, u64 q, u64 r0, u64 r1) {
  u64 v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (threadIdx.x * 977 + i * 13 + 5) % q;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = reduce128((u128)v[i] * (v[(i + 1) % ILP] | 1), q, r0, r1);
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// D: one-word Barrett: x < 2^(2b); xs = x >> (b-2); qhat = hi64(xs * mu), mu = floor(2^(b+62)/q); r in [0,3q)
__device__ __forceinline__ u64 reduce_b1(u128 x, u64 q, u64 mu, int sh) {
  const u64 xs = (u64)(x >> sh);
  const u64 qhat = mulhi64(xs, mu);
  u64 r = (u64)x - qhat * q;
  r = r >= 2 * q ? r - 2 * q : r;
  return r >= q ? r - q : r;
}
__global__ void k_mul_barrett1(u64* out, u64 q, u64 mu, int sh) {
  u64 v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (threadIdx.x * 977 + i * 13 + 5) % q;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = reduce_b1((u128)v[i] * (v[(i + 1) % ILP] | 1), q, mu, sh);
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// E: FP64 variable x variable modmul for q < 2^50 (values as doubles)
__global__ void k_mul_f64(double* out, double q, double qinv) {
  double v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (double)((threadIdx.x * 977 + i * 13 + 5) % 1000000);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      double a = v[i], b = v[(i + 1) % ILP];
      double xh = a * b, xl = fma(a, b, -xh);
      double qf = rint(xh * qinv);
      v[i] = fma(-qf, q, xh) + xl;
    }
  }
  double s = 0; for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// F: pseudo-Mersenne reduction for q = 2^61 - c (c < 2^24): x*y mod q
__device__ __forceinline__ u64 mul_pm61(u64 a, u64 b, u64 q, u32 c) {
  u128 p = (u128)a * b;                       // < 2^122 (+ lazy slack)
  u64 lo = (u64)p & ((1ull << 61) - 1);
  u64 hi = (u64)(p >> 61);                    // < 2^63
  u128 t = (u128)hi * c + lo;                 // < 2^87
  u64 lo2 = (u64)t & ((1ull << 61) - 1);
  u64 hi2 = (u64)(t >> 61);                   // < 2^26
  u64 r = lo2 + hi2 * c;                      // < 2^61 + 2^50
  return r >= q ? r - q : r;
}
__global__ void k_mul_pm61(u64* out, u64 q, u32 c) {
  u64 v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (threadIdx.x * 977 + i * 13 + 5) % q;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = mul_pm61(v[i], v[(i + 1) % ILP] | 1, q, c);
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// G: butterfly with pseudo-Mersenne twiddle multiply (no Shoup quotient), q = 2^61 - c
__global__ void k_bfly_pm61(u64* out, const u64* tw, u64 q, u32 c) {
  u64 v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (threadIdx.x * 977 + i * 13) % q;
  for (int it = 0; it < ITERS; it++) {
    const u64 w = tw[2 * (it & 63)];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      u64 &X = v[2 * i], &Y = v[2 * i + 1];
      u64 t = mul_pm61(Y, w, q, c);   // canonical
      u64 x = X;
      u64 s = x + t; X = s >= q ? s - q : s;
      Y = x >= t ? x - t : x + q - t;
    }
  }
  u64 s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  void* buf; hipMalloc(&buf, (size_t)blocks * threads * 8);
  const double lanes = (double)blocks * threads;
  auto report = [&](const char* name, float ms, double ops_per_lane) {
    double tops = lanes * ops_per_lane / (ms * 1e-3) / 1e12;
    printf("%-18s %8.3f ms  %8.3f Tops/s  (%.2f /clk/CU @2.4GHz => %.1f slots each)\n", name, ms, tops, tops * 1e12 / (p.multiProcessorCount * 2.4e9),
           107.0 / (tops * 1e12 / (p.multiProcessorCount * 2.4e9)));
  };
  const u64 q44 = 0xfffffffc001ull, q61 = 0x1ffffffffff0c001ull;
  u64 htw[128]; double hdw[128];
  for (int i = 0; i < 64; i++) {
    u64 w = (123456789123ull * (i + 3)) % q44;
    htw[2 * i] = w; htw[2 * i + 1] = (u64)(((u128)w << 64) / q44);
    hdw[2 * i] = (double)w; hdw[2 * i + 1] = (double)w / (double)q44;
  }
  u64* dtw; double* ddw; hipMalloc(&dtw, sizeof(htw)); hipMalloc(&ddw, sizeof(hdw));
  hipMemcpy(dtw, htw, sizeof(htw), hipMemcpyHostToDevice); hipMemcpy(ddw, hdw, sizeof(hdw), hipMemcpyHostToDevice);
  report("bfly_int", time_ms([&] { k_bfly_int<<<blocks, threads>>>((u64*)buf, dtw, q44); }), (double)ITERS * ILP);
  report("bfly_f64+red/4", time_ms([&] { k_bfly_f64<<<blocks, threads>>>((double*)buf, ddw, (double)q44); }), (double)ITERS * ILP);
  report("bfly_f64", time_ms([&] { k_bfly_f64_nored<<<blocks, threads>>>((double*)buf, ddw, (double)q44); }), (double)ITERS * ILP);
  report("bfly_pm61", time_ms([&] { k_bfly_pm61<<<blocks, threads>>>((u64*)buf, dtw, q61, (u32)((1ull << 61) - q61)); }), (double)ITERS * ILP);
  {
    u128 R = (~(u128)0) / q61;
    report("mul_barrett2w", time_ms([&] { k_mul_barrett2<<<blocks, threads>>>((u64*)buf, q61, (u64)R, (u64)(R >> 64)); }), (double)ITERS * ILP);
    int b = 61; u64 mu = (u64)((((u128)1) << (b + 62)) / q61);
    report("mul_barrett1w", time_ms([&] { k_mul_barrett1<<<blocks, threads>>>((u64*)buf, q61, mu, b - 2); }), (double)ITERS * ILP);
    report("mul_pm61", time_ms([&] { k_mul_pm61<<<blocks, threads>>>((u64*)buf, q61, (u32)((1ull << 61) - q61)); }), (double)ITERS * ILP);
    report("mul_f64(q44)", time_ms([&] { k_mul_f64<<<blocks, threads>>>((double*)buf, (double)q44, 1.0 / (double)q44); }), (double)ITERS * ILP);
  }
  return 0;
}
