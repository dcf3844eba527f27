// Probe (developer tool): do HIP events recorded INSIDE a captured graph time its kernels on replay?
// Build: hipcc --offload-arch=gfx950 -O3 tools/graph_event_probe.hip -o build/graph_event_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(double *p, int iters) {
    double v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
    p[threadIdx.x] = v;
}
int main() {
    double *p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e[4]; for (auto &x : e) CK(hipEventCreate(&x));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(e[0], st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, p, 200000);
    CK(hipEventRecord(e[1], st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, p, 800000);
    CK(hipEventRecord(e[2], st));
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float a = -1, b = -1;
        hipError_t ea = hipEventElapsedTime(&a, e[0], e[1]), eb = hipEventElapsedTime(&b, e[1], e[2]);
        printf("replay %d: kernel 1 %.3f ms (%s), kernel 2 %.3f ms (%s)\n", rep, a, hipGetErrorString(ea), b, hipGetErrorString(eb));
    }
    // eager reference
    CK(hipEventRecord(e[0], st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, p, 200000);
    CK(hipEventRecord(e[1], st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, p, 800000);
    CK(hipEventRecord(e[2], st));
    CK(hipStreamSynchronize(st));
    float a, b; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2]));
    printf("eager:    kernel 1 %.3f ms, kernel 2 %.3f ms\n", a, b);
    return 0;
}
