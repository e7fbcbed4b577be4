// Device-copy peak of this box's HBM ("replace nominal with a measured device-copy peak on the box").
// Grid-stride float4 copy, 256 CUs x 8 workgroups x 256 threads, 16 B per lane per access (coalesced 1 KB per wave);
// bytes counted = read + written.  Also a read-only pass (sum into a register, one store per thread) and hipMemcpyDtoD.
//   hipcc --offload-arch=gfx950 -O3 devcopy.hip -o devcopy && ./devcopy
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void __launch_bounds__(256) copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read_kernel(const float4 *__restrict__ src, float *__restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b;
    float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 2048 * 256 * 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    for (int grid : {1024, 2048, 4096, 8192}) {
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy  grid %5d: %.1f GB/s (read + write, 2 GiB buffers)\n", grid, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
    }
    for (int grid : {2048}) {
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, o, n);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, o, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read  grid %5d: %.1f GB/s (read only)\n", grid, 1.0 * bytes * reps / (ms * 1e-3) / 1e9);
    }
    hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyDtoD : %.1f GB/s (read + write)\n", 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
    return 0;
}
