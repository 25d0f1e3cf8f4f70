// graph_gap.hip -- time per step of a chain of 6 dependent ~20 us kernels: plain launches vs a replayed hipGraph
// (with and without updating every kernel node's parameters before each replay).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { float *p; int n; int pad[96]; };   // ~400-byte by-value argument like BevArgs
__global__ void k(const Args a)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) a.p[i] = a.p[i] * 1.0001f + 1.0f;
}
int main()
{
    const int n = 64 << 20;
    Args a{}; hipMalloc(&a.p, (size_t)n * 4); a.n = n; hipMemset(a.p, 0, (size_t)n * 4);
    hipStream_t s; hipStreamCreate(&s);
    const int K = 6, STEPS = 300;
    auto plain = [&]() { for (int j = 0; j < K; ++j) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, s, a); };
    for (int i = 0; i < 20; ++i) plain();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < STEPS; ++i) plain();
    hipStreamSynchronize(s);
    double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / STEPS;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); plain(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < STEPS; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / STEPS;
    // with parameter updates
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    hipGraphNode_t nodes[16]; hipGraphGetNodes(g, nodes, &nn);
    void *kargs[] = {&a};
    hipKernelNodeParams kp{}; kp.func = (void *)k; kp.gridDim = dim3(2048); kp.blockDim = dim3(256); kp.kernelParams = kargs;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < STEPS; ++i) {
        for (size_t j = 0; j < nn; ++j) hipGraphExecKernelNodeSetParams(ge, nodes[j], &kp);
        hipGraphLaunch(ge, s);
    }
    hipStreamSynchronize(s);
    double us_graph_upd = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / STEPS;
    // one kernel alone, to know the pure kernel time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, s, a); hipEventRecord(e1, s);
    hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chain of %d kernels: plain %.1f us/step, graph %.1f us/step, graph + %zu param updates %.1f us/step (kernel alone ~%.1f us)\n",
           K, us_plain, us_graph, nn, us_graph_upd, ms * 1000 / 50);
    return 0;
}
