// Issue rates of the individual FP64 instructions of the ArithD butterfly on gfx950, and of the butterfly itself
// at several ILP / occupancy points.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb3 tools/microbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>

constexpr int ITERS = 2048;

template <int OP, int ILP>
__global__ void k_op(double* out, double a, double b) {
  double v[ILP];
  for (int i = 0; i < ILP; i++) v[i] = (double)(threadIdx.x + i) * 1e-3 + 1.0;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) v[i] = fma(v[i], a, b);
      if (OP == 1) v[i] = v[i] * a;
      if (OP == 2) v[i] = v[i] + a;
      if (OP == 3) v[i] = rint(v[i]) ;
      if (OP == 4) v[i] = fma(v[i], v[(i + 1) % ILP], v[(i + 2) % ILP]);  // three distinct register operands
      if (OP == 5) v[i] = v[i] * v[(i + 1) % ILP];
      if (OP == 6) v[i] = v[i] + v[(i + 1) % ILP];
    }
    if (OP == 3) {
#pragma unroll
      for (int i = 0; i < ILP; i++) asm volatile("" : "+v"(v[i]));
    }
  }
  double s = 0;
  for (int i = 0; i < ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, bool MAGIC>
__global__ void k_bfly(double* out, const double* tw, double q) {
  double v[2 * ILP];
  for (int i = 0; i < 2 * ILP; i++) v[i] = (double)((threadIdx.x * 977 + i * 13) % 100000);
  for (int it = 0; it < ITERS; it++) {
    const double w = tw[2 * (it & 63)], wq = tw[2 * (it & 63) + 1];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      double &X = v[2 * i], &Y = v[2 * i + 1];
      double qf;
      if (MAGIC) {
        const double M = 6755399441055744.0;  // 1.5 * 2^52
        qf = fma(Y, wq, M) - M;
      } else {
        qf = rint(Y * wq);
      }
      double xh = Y * w;
      double xl = fma(Y, w, -xh);
      double t = fma(-qf, q, xh) + xl;
      double x = X;
      X = x + t;
      Y = x - t;
    }
    if ((it & 7) == 7) {
#pragma unroll
      for (int i = 0; i < 2 * ILP; i++) v[i] = fma(-rint(v[i] * (1.0 / q)), q, v[i]);
    }
  }
  double s = 0;
  for (int i = 0; i < 2 * ILP; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float time_ms(const std::function<void()>& f) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;
  double* buf;
  hipMalloc(&buf, 1 << 26);
  double htw[128];
  const double q = 17592186028033.0;
  for (int i = 0; i < 64; i++) {
    htw[2 * i] = (double)(123456789ull * (i + 1) % 17592186028033ull);
    htw[2 * i + 1] = htw[2 * i] / q;
  }
  double* dtw;
  hipMalloc(&dtw, sizeof(htw));
  hipMemcpy(dtw, htw, sizeof(htw), hipMemcpyHostToDevice);
  auto rep = [&](const char* name, float ms, double ops_per_thread, int blocks, int threads) {
    const double total = ops_per_thread * blocks * threads;
    printf("%-28s blocks/CU=%d thr=%d  %8.3f ms  %7.2f lane-ops/clk/CU\n", name, blocks / cus, threads, ms, total / (ms * 1e-3) / clk / cus);
  };
  for (int wpc : {4, 8, 16}) {  // waves per CU = blocks/CU * 4
    const int blocks = cus * wpc / 4, threads = 256;
#define OPB(OP, NAME) rep(NAME, time_ms([&] { k_op<OP, 8><<<blocks, threads>>>(buf, 1.0000001, 1e-9); }), (double)ITERS * 8, blocks, threads);
    OPB(0, "fma(v,const,const)")
    OPB(4, "fma(v,v,v)")
    OPB(1, "mul(v,const)")
    OPB(5, "mul(v,v)")
    OPB(2, "add(v,const)")
    OPB(6, "add(v,v)")
    OPB(3, "rndne(v)")
    rep("bfly ILP2 rint", time_ms([&] { k_bfly<2, false><<<blocks, threads>>>(buf, dtw, q); }), (double)ITERS * 2, blocks, threads);
    rep("bfly ILP4 rint", time_ms([&] { k_bfly<4, false><<<blocks, threads>>>(buf, dtw, q); }), (double)ITERS * 4, blocks, threads);
    rep("bfly ILP8 rint", time_ms([&] { k_bfly<8, false><<<blocks, threads>>>(buf, dtw, q); }), (double)ITERS * 8, blocks, threads);
    rep("bfly ILP4 magic", time_ms([&] { k_bfly<4, true><<<blocks, threads>>>(buf, dtw, q); }), (double)ITERS * 4, blocks, threads);
    rep("bfly ILP8 magic", time_ms([&] { k_bfly<8, true><<<blocks, threads>>>(buf, dtw, q); }), (double)ITERS * 8, blocks, threads);
  }
  return 0;
}
