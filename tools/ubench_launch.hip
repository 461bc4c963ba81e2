// The floor under a one-frame call: N dependent empty kernels on one stream + hipStreamSynchronize, eager and as a replayed
// graph.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_launch tools/ubench_launch.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_spin(long long cycles, int* p) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 9999) *p = 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    int* d;
    hipMalloc(&d, 4);
    for (int n : {1, 2, 3, 5, 8}) {
        for (int rep = 0; rep < 50; ++rep) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(22), dim3(256), 0, s, d); hipStreamSynchronize(s); }
        const int R = 500;
        double t0 = now();
        for (int rep = 0; rep < R; ++rep) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(22), dim3(256), 0, s, d); hipStreamSynchronize(s); }
        const double eager = (now() - t0) / R;
        // 10 us of work per kernel (100 MHz wall clock): does the launch cost hide behind the previous kernel?
        t0 = now();
        for (int rep = 0; rep < R; ++rep) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(22), dim3(256), 0, s, 1000LL, d); hipStreamSynchronize(s); }
        const double spin = (now() - t0) / R;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(22), dim3(256), 0, s, d);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 50; ++rep) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
        t0 = now();
        for (int rep = 0; rep < R; ++rep) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
        const double graph = (now() - t0) / R;
        printf("%d dependent empty kernels + synchronise: eager %.1f us, graph replay %.1f us; with 10 us of work each: %.1f us (= %.1f over the work)\n",
               n, eager, graph, spin, spin - 10.0 * n);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
