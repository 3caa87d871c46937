// Microbenchmark behind DESIGN 3.3: what HBM gives a lane-per-record kernel that walks D column streams of a column-major
// [D][B] fp32 matrix, as a function of the bytes a lane moves per instruction (V consecutive records per lane: 4, 8 or 16 B).
// Build: hipcc --offload-arch=gfx950 -O3 -o benchmarks/micro/colstreams benchmarks/micro/colstreams.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int V> struct Vec { float v[V]; };

template <int V>
__global__ __launch_bounds__(256) void k_write(float* out, long B, int D) {
    const long r0 = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    if (r0 >= B) return;
    Vec<V> x;
#pragma unroll
    for (int i = 0; i < V; ++i) x.v[i] = (float)(r0 + i);
    for (int c = 0; c < D; ++c) *reinterpret_cast<Vec<V>*>(out + (long)c * B + r0) = x;
}
template <int V>
__global__ __launch_bounds__(256) void k_read(const float* in, float* sum, long B, int D) {
    const long r0 = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    if (r0 >= B) return;
    float acc = 0.0f;
    for (int c0 = 0; c0 < D; c0 += 8) {  // eight loads in flight
        Vec<V> x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const Vec<V>*>(in + (long)(c0 + j < D ? c0 + j : D - 1) * B + r0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) acc += x[j].v[i];
    }
    if (acc == 1.2345f) sum[0] = acc;
}
template <int V>
__global__ __launch_bounds__(256) void k_copy(const float* in, float* out, long B, int D) {
    const long r0 = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    if (r0 >= B) return;
    for (int c0 = 0; c0 < D; c0 += 8) {
        Vec<V> x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const Vec<V>*>(in + (long)(c0 + j < D ? c0 + j : D - 1) * B + r0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < D) *reinterpret_cast<Vec<V>*>(out + (long)(c0 + j) * B + r0) = x[j];
    }
}

template <class F>
double timed(F&& launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20.0;
}

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : (1L << 20);
    const int D = argc > 2 ? atoi(argv[2]) : 120;
    float *in, *out, *sum;
    hipMalloc(&in, sizeof(float) * B * D);
    hipMalloc(&out, sizeof(float) * B * D);
    hipMalloc(&sum, 4);
    hipMemset(in, 0, sizeof(float) * B * D);
    const double bytes = (double)B * D * 4;
#define RUN(V)                                                                                                                     \
    {                                                                                                                              \
        const int grid = (int)((B / V + 255) / 256);                                                                               \
        const double w = timed([&] { hipLaunchKernelGGL(k_write<V>, dim3(grid), dim3(256), 0, 0, out, B, D); });                   \
        const double r = timed([&] { hipLaunchKernelGGL(k_read<V>, dim3(grid), dim3(256), 0, 0, in, sum, B, D); });                \
        const double c = timed([&] { hipLaunchKernelGGL(k_copy<V>, dim3(grid), dim3(256), 0, 0, in, out, B, D); });                \
        printf("{\"rows\": %ld, \"columns\": %d, \"bytes_per_lane\": %d, \"write_GBs\": %.0f, \"read_GBs\": %.0f, \"copy_GBs\": %.0f}\n", B, D, 4 * V, \
               bytes / w / 1e6, bytes / r / 1e6, 2 * bytes / c / 1e6);                                                             \
    }
    RUN(1) RUN(2) RUN(4)
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
