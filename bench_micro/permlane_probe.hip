// Developer probe: semantics of v_permlane32_swap / v_permlane16_swap on gfx950 (which lanes of which operand move).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    u2 s = __builtin_amdgcn_permlane16_swap(1000u + l, 2000u + l, false, false);
    o[l] = r.x; o[64 + l] = r.y; o[128 + l] = s.x; o[192 + l] = s.y;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[4] = {"swap32.x", "swap32.y", "swap16.x", "swap16.y"};
    for (int i = 0; i < 4; ++i) {
        printf("%s:", nm[i]);
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * i + l]);
        printf("\n");
    }
    return 0;
}
