// tools/microbench_hbm.hip -- what HBM delivers to plain kernels on this box (the denominators behind DESIGN.md's "store roof"
// and the stand-alone NTT's 4.8 TB/s): read-only, write-only and copy kernels over 4 GiB, in the access shapes the library's
// kernels use (16-byte and 8-byte accesses, grid-stride and one-contiguous-chunk-per-workgroup, plain and non-temporal).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_hbm tools/microbench_hbm.hip && /tmp/mb_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <bool NT, class T>
__device__ __forceinline__ T ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void st(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// grid-stride over n elements of T
template <class T, bool NT>
__global__ void k_copy(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) st<NT>(out + i, ld<NT>(in + i));
}
template <class T, bool NT>
__global__ void k_read(const T* __restrict__ in, u64* __restrict__ sink, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  u64 acc = 0;
  for (; i < n; i += stride) {
    const T v = ld<NT>(in + i);
    if constexpr (sizeof(T) == 16) acc ^= v.x ^ v.y;
    else acc ^= (u64)v;
  }
  if (acc == 0x123456789abcdefull) sink[0] = acc;  // (never true: keeps the loads)
}
template <class T, bool NT>
__global__ void k_write(T* __restrict__ out, size_t n, u64 seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  T v;
  if constexpr (sizeof(T) == 16) v = T{seed, seed + 1};
  else v = (T)seed;
  for (; i < n; i += stride) st<NT>(out + i, v);
}
// the stand-alone transform's shape: one workgroup per contiguous CHUNK (64 KB = a polynomial of N = 8192): every element is
// loaded into registers first (16 per thread at 512 threads), then everything is stored -- to `out` (copy) or back in place
template <int EPT, bool NT>
__global__ void k_chunk(const u64* __restrict__ in, u64* __restrict__ out, int rounds) {
  const size_t chunk = (size_t)blockDim.x * EPT;
  const u64* src = in + (size_t)blockIdx.x * chunk;
  u64* dst = out + (size_t)blockIdx.x * chunk;
  u64 v[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) v[e] = ld<NT>(src + (size_t)e * blockDim.x + threadIdx.x);
  for (int r = 0; r < rounds; r++) {  // (rounds = 0: a pure copy; > 0: a dependent ALU chain between the loads and the stores)
#pragma unroll
    for (int e = 0; e < EPT; e++) v[e] = v[e] * 0x9e3779b97f4a7c15ull + v[(e + 1) % EPT];
  }
#pragma unroll
  for (int e = 0; e < EPT; e++) st<NT>(dst + (size_t)e * blockDim.x + threadIdx.x, v[e]);
}

// the stand-alone INVERSE transform's shape: every thread loads runs of 2^RF consecutive elements with 16-byte loads (a wavefront's load
// instruction covers 64 x 16 bytes at a stride of 8 << RF), and stores 8-byte words, lanes contiguous
template <int EPT, int RF>
__global__ void k_chunk_runs(const u64* __restrict__ in, u64* __restrict__ out) {
  const size_t chunk = (size_t)blockDim.x * EPT;
  const u64* src = in + (size_t)blockIdx.x * chunk;
  u64* dst = out + (size_t)blockIdx.x * chunk;
  u64 v[EPT];
#pragma unroll
  for (int g = 0; g < (EPT >> RF); g++) {
    const u64x2* p = reinterpret_cast<const u64x2*>(src + ((size_t)(threadIdx.x + g * blockDim.x) << RF));
#pragma unroll
    for (int k = 0; k < (1 << RF); k += 2) {
      const u64x2 w = p[k >> 1];
      v[g * (1 << RF) + k] = w.x, v[g * (1 << RF) + k + 1] = w.y;
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; e++) dst[(size_t)e * blockDim.x + threadIdx.x] = v[e];
}
// ... and the mirror image: 8-byte loads, lanes contiguous; runs of 2^RF stored with 16-byte stores
template <int EPT, int RF>
__global__ void k_chunk_run_stores(const u64* __restrict__ in, u64* __restrict__ out) {
  const size_t chunk = (size_t)blockDim.x * EPT;
  const u64* src = in + (size_t)blockIdx.x * chunk;
  u64* dst = out + (size_t)blockIdx.x * chunk;
  u64 v[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) v[e] = src[(size_t)e * blockDim.x + threadIdx.x];
#pragma unroll
  for (int g = 0; g < (EPT >> RF); g++) {
    u64x2* p = reinterpret_cast<u64x2*>(dst + ((size_t)(threadIdx.x + g * blockDim.x) << RF));
#pragma unroll
    for (int k = 0; k < (1 << RF); k += 2) p[k >> 1] = u64x2{v[g * (1 << RF) + k], v[g * (1 << RF) + k + 1]};
  }
}

template <typename F>
static float time_ms(F f, int reps = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
  const size_t bytes = (size_t)4 << 30;
  void *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMemset(dst, 0, bytes));
  const int cus = p.multiProcessorCount;
  auto tb = [&](double moved, float ms) { return moved / (ms * 1e-3) / 1e12; };
  for (int wg_per_cu : {4, 8, 16, 32}) {
    const int g = cus * wg_per_cu;
    float ms;
    ms = time_ms([&] { k_read<u64x2, false><<<g, 256>>>((const u64x2*)src, (u64*)dst, bytes / 16); });
    printf("read   16B        grid %5d x256: %7.3f ms  %5.2f TB/s\n", g, ms, tb((double)bytes, ms));
    ms = time_ms([&] { k_write<u64x2, false><<<g, 256>>>((u64x2*)dst, bytes / 16, 7); });
    printf("write  16B        grid %5d x256: %7.3f ms  %5.2f TB/s\n", g, ms, tb((double)bytes, ms));
    ms = time_ms([&] { k_write<u64x2, true><<<g, 256>>>((u64x2*)dst, bytes / 16, 7); });
    printf("write  16B nt     grid %5d x256: %7.3f ms  %5.2f TB/s\n", g, ms, tb((double)bytes, ms));
    ms = time_ms([&] { k_write<u64, false><<<g, 256>>>((u64*)dst, bytes / 8, 7); });
    printf("write   8B        grid %5d x256: %7.3f ms  %5.2f TB/s\n", g, ms, tb((double)bytes, ms));
    ms = time_ms([&] { k_copy<u64x2, false><<<g, 256>>>((const u64x2*)src, (u64x2*)dst, bytes / 16); });
    printf("copy   16B        grid %5d x256: %7.3f ms  %5.2f TB/s (read + write)\n", g, ms, tb(2.0 * bytes, ms));
    ms = time_ms([&] { k_copy<u64x2, true><<<g, 256>>>((const u64x2*)src, (u64x2*)dst, bytes / 16); });
    printf("copy   16B nt     grid %5d x256: %7.3f ms  %5.2f TB/s (read + write)\n", g, ms, tb(2.0 * bytes, ms));
    ms = time_ms([&] { k_copy<u64, false><<<g, 256>>>((const u64*)src, (u64*)dst, bytes / 8); });
    printf("copy    8B        grid %5d x256: %7.3f ms  %5.2f TB/s (read + write)\n", g, ms, tb(2.0 * bytes, ms));
  }
  // one 64 KB chunk per 512-thread workgroup (16 elements per thread), 4 GiB = 65536 chunks
  {
    const int chunks = (int)(bytes / (512 * 16 * 8));
    for (int rounds : {0, 8, 32}) {
      float ms = time_ms([&] { k_chunk<16, false><<<chunks, 512>>>((const u64*)src, (u64*)dst, rounds); });
      printf("chunk 64KB copy      rounds %2d: %7.3f ms  %5.2f TB/s (read + write)\n", rounds, ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk<16, true><<<chunks, 512>>>((const u64*)src, (u64*)dst, rounds); });
      printf("chunk 64KB copy nt   rounds %2d: %7.3f ms  %5.2f TB/s (read + write)\n", rounds, ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk<16, false><<<chunks, 512>>>((const u64*)dst, (u64*)dst, rounds); });
      printf("chunk 64KB in place  rounds %2d: %7.3f ms  %5.2f TB/s (read + write)\n", rounds, ms, tb(2.0 * bytes, ms));
    }
    {
      float ms = time_ms([&] { k_chunk_runs<16, 1><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, loads in runs of 2 : %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk_runs<16, 2><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, loads in runs of 4 : %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk_runs<16, 3><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, loads in runs of 8 : %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk_runs<16, 4><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, loads in runs of 16: %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk_run_stores<16, 2><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, stores in runs of 4: %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
      ms = time_ms([&] { k_chunk_run_stores<16, 4><<<chunks, 512>>>((const u64*)src, (u64*)dst); });
      printf("chunk 64KB, stores in runs of 16: %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
    }
    // 32 KB chunks at 256 threads (the edge kernels' workgroup size)
    const int chunks2 = (int)(bytes / (256 * 16 * 8));
    float ms = time_ms([&] { k_chunk<16, false><<<chunks2, 256>>>((const u64*)src, (u64*)dst, 0); });
    printf("chunk 32KB copy (256 thr)     : %7.3f ms  %5.2f TB/s (read + write)\n", ms, tb(2.0 * bytes, ms));
  }
  return 0;
}
