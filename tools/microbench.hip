// tools/microbench.hip -- instruction-rate probes for gfx950 that drive the kernel design in DESIGN.md:
// 64-bit modular butterflies (integer Shoup vs FP64-assisted) and a plain HBM copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

constexpr int ITERS = 2048;
constexpr int ILP = 8;

// ---- raw instruction probes -------------------------------------------------
__global__ void k_mad64(u64* out, u32 a, u32 b) {
  u64 acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (u64)((u32)acc[i]) * (u64)b + acc[i];  // v_mad_u64_u32
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo32(u32* out, u32 b) {
  u32 acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] * b;  // v_mul_lo_u32
  }
  u32 s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi32(u32* out, u32 b) {
  u32 acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = 0x80000000u + threadIdx.x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i], b) | 0x80000000u;  // v_mul_hi_u32 + v_or
  }
  u32 s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(u32* out, u32 b) {
  u32 acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + b) ^ (u32)it;  // 2 simple ops
  }
  u32 s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(u64* out, u64 b) {
  u64 acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] + b;  // v_lshl_add_u64 or add_co/addc
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(double* out, double b, double c) {
  double acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = fma(acc[i], b, c);  // v_fma_f64
  }
  double s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rnd64(double* out, double b) {
  double acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x * 1.37 + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = rint(acc[i]) + b;  // v_rndne_f64 + v_add_f64
  }
  double s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma32(float* out, float b, float c) {
  float acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = fmaf(acc[i], b, c);
  }
  float s = 0; for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- butterflies -------------------------------------------------------------
// Harvey/Shoup lazy butterfly on u64: X,Y in [0,4q)
__device__ __forceinline__ void bfly_int(u64& X, u64& Y, u64 w, u64 wq, u64 q, u64 q2) {
  u64 x = X >= q2 ? X - q2 : X;
  u64 h = mulhi64(Y, wq);
  u64 t = Y * w - h * q;
  X = x + t;
  Y = x + q2 - t;
}
__global__ void k_bfly_int(u64* out, u64 w, u64 wq, u64 q) {
  u64 v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (threadIdx.x * 977 + i * 13) % q;
  u64 q2 = q << 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) bfly_int(v[2 * i], v[2 * i + 1], w + it, wq + it, q, q2);
  }
  u64 s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// FP64-assisted signed butterfly for q < 2^49: values are exact integers held in doubles.
__device__ __forceinline__ void bfly_f64(double& X, double& Y, double w, double wq /* w/q */, double q) {
  double qf = rint(Y * wq);
  double xh = Y * w;
  double xl = fma(Y, w, -xh);
  double t = fma(-qf, q, xh) + xl;
  double x = X;
  X = x + t;
  Y = x - t;
}
__global__ void k_bfly_f64(double* out, double w, double wq, double q) {
  double v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (double)((threadIdx.x * 977 + i * 13) % 1000);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) bfly_f64(v[2 * i], v[2 * i + 1], w, wq, q);
#pragma unroll
    for (int i = 0; i < 2 * ILP; i++) v[i] = v[i] - q * rint(v[i] * (1.0 / 17592186044423.0)) * 0.0;  // keep bounded (compiled out mostly)
  }
  double s = 0; for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Barrett 128->64 modular multiply (dyadic product), u64
__device__ __forceinline__ u64 mulmod_barrett(u64 a, u64 b, u64 q, u64 r0, u64 r1) {
  u64 lo = a * b, hi = mulhi64(a, b);
  u64 carry = mulhi64(lo, r0);
  u64 t2lo = lo * r1, t2hi = mulhi64(lo, r1);
  u64 t1 = t2lo + carry; u64 t3 = t2hi + (t1 < t2lo);
  u64 t4lo = hi * r0, t4hi = mulhi64(hi, r0);
  u64 t5 = t1 + t4lo; carry = t4hi + (t5 < t1);
  u64 qh = hi * r1 + t3 + carry;
  u64 r = lo - qh * q;
  return r >= q ? r - q : r;
}
__global__ void k_barrett(u64* out, u64 q, u64 r0, u64 r1) {
  u64 v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (threadIdx.x * 977 + i * 13 + 5) % q;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = mulmod_barrett(v[i], v[(i + 1) % ILP] | 1, q, r0, r1);
  }
  u64 s = 0; for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
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
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  void* buf; CK(hipMalloc(&buf, (size_t)blocks * threads * 8));
  const double lanes = (double)blocks * threads;
  auto report = [&](const char* name, float ms, double ops_per_lane) {
    double tops = lanes * ops_per_lane / (ms * 1e-3) / 1e12;
    // lane-ops/clk/CU at 2.4 GHz
    printf("%-14s %8.3f ms  %8.3f Tops/s  (%.1f lane-ops/clk/CU @2.4GHz)\n", name, ms, tops, tops * 1e12 / (p.multiProcessorCount * 2.4e9));
  };
  const u64 q = 0xffffffffc001ull /*44+ bit*/, q61 = 0x1ffffffffff0c001ull;
  report("mad_u64_u32", time_ms([&] { k_mad64<<<blocks, threads>>>((u64*)buf, 3, 12345); }), (double)ITERS * ILP);
  report("mul_lo_u32", time_ms([&] { k_mullo32<<<blocks, threads>>>((u32*)buf, 12345); }), (double)ITERS * ILP);
  report("mul_hi_u32+or", time_ms([&] { k_mulhi32<<<blocks, threads>>>((u32*)buf, 0xfffffff1u); }), (double)ITERS * ILP);
  report("add32+xor", time_ms([&] { k_add32<<<blocks, threads>>>((u32*)buf, 12345); }), (double)ITERS * ILP * 2);
  report("add64", time_ms([&] { k_add64<<<blocks, threads>>>((u64*)buf, 0x123456789abcull); }), (double)ITERS * ILP);
  report("fma_f64", time_ms([&] { k_fma64<<<blocks, threads>>>((double*)buf, 1.0000001, 0.5); }), (double)ITERS * ILP);
  report("rndne+add f64", time_ms([&] { k_rnd64<<<blocks, threads>>>((double*)buf, 0.37); }), (double)ITERS * ILP * 2);
  report("fma_f32", time_ms([&] { k_fma32<<<blocks, threads>>>((float*)buf, 1.0000001f, 0.5f); }), (double)ITERS * ILP);
  report("bfly_int44", time_ms([&] { k_bfly_int<<<blocks, threads>>>((u64*)buf, 123456789123ull, (u64)(((unsigned __int128)123456789123ull << 64) / q), q); }), (double)ITERS * ILP);
  report("bfly_int61", time_ms([&] { k_bfly_int<<<blocks, threads>>>((u64*)buf, 123456789123ull, (u64)(((unsigned __int128)123456789123ull << 64) / q61), q61); }), (double)ITERS * ILP);
  report("bfly_f64", time_ms([&] { k_bfly_f64<<<blocks, threads>>>((double*)buf, 123456789123.0, 123456789123.0 / (double)q, (double)q); }), (double)ITERS * ILP);
  {
    unsigned __int128 R = (~(unsigned __int128)0) / q61;
    report("barrett61", time_ms([&] { k_barrett<<<blocks, threads>>>((u64*)buf, q61, (u64)R, (u64)(R >> 64)); }), (double)ITERS * ILP);
  }
  // HBM copy: 2 GiB in, 2 GiB out
  size_t bytes = (size_t)2 << 30;
  void *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  float ms = time_ms([&] { k_copy<<<p.multiProcessorCount * 16, 256>>>((const uint4*)src, (uint4*)dst, bytes / 16); });
  printf("copy 2GiB: %.3f ms -> %.2f TB/s (read+write)\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
  return 0;
}
