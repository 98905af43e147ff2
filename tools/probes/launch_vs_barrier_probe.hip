// What does a dependent stage cost on this part WITHOUT its arithmetic?  (is a persistent depth-decoder kernel with
// grid barriers cheaper than the captured graph's kernel boundaries?)
//   (a) a HIP graph of N dependent launches of a kernel that only loads a 2-KB vector another launch wrote and stores one back
//       (the hand-off of a decode stage: 256 workgroups x 256 threads, the shape of the decoder's matrix-vector launches);
//   (b) ONE launch of 256 workgroups (one per CU) running the same N hand-offs separated by a grid barrier: every workgroup adds
//       to an agent-scope counter after its stores are out (sc1, drained), one lane polls it with sc1 loads, then a workgroup
//       barrier - the recipe of the micro-architecture guide (barrier-counter / XCD-hierarchical form).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/launch_vs_barrier_probe.hip -o /tmp/lvb && /tmp/lvb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void stage_kernel(const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float xs[512];
    for (int i = threadIdx.x; i < 512; i += 256) xs[i] = in[i];
    __syncthreads();
    float a = 0.f;
    for (int i = 0; i < 512; i += 64) a += xs[(i + threadIdx.x) & 511];
    if (threadIdx.x < 2) out[blockIdx.x * 2 + threadIdx.x] = a * 1e-3f;      // 256 workgroups x 2 floats = the next 2-KB vector
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();                                  // every wave's stores issued
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void persistent_kernel(float* __restrict__ a, float* __restrict__ b, unsigned* counter, int n) {
    __shared__ float xs[512];
    for (int s = 0; s < n; ++s) {
        const float* in = (s & 1) ? b : a;
        float* out = (s & 1) ? a : b;
        for (int i = threadIdx.x; i < 512; i += 256) xs[i] = __builtin_nontemporal_load(in + i);
        __syncthreads();
        float acc = 0.f;
        for (int i = 0; i < 512; i += 64) acc += xs[(i + threadIdx.x) & 511];
        if (threadIdx.x < 2) __builtin_nontemporal_store(acc * 1e-3f, out + blockIdx.x * 2 + threadIdx.x);
        grid_barrier(counter, (unsigned)(s + 1) * gridDim.x);
    }
}

int main() {
    const int N = 600;
    float *a, *b;
    unsigned* counter;
    CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&counter, 4));
    CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < N; ++s) hipLaunchKernelGGL(stage_kernel, dim3(256), dim3(256), 0, st, (s & 1) ? b : a, (s & 1) ? a : b);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) graph of %d dependent 256-workgroup launches: %.2f us per stage\n", N, ms * 1e3 / (10.0 * N));
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, st));
        hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(256), 0, st, a, b, counter, N); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) { CK(hipMemsetAsync(counter, 0, 4, st)); hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(256), 0, st, a, b, counter, N); }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) one persistent launch, %d hand-offs separated by a grid barrier (256 workgroups): %.2f us per stage\n", N, ms * 1e3 / (10.0 * N));
    }
    return 0;
}
