// Does the output clamp of the packed float32 add work on this chip?  v_pk_add_f32 ... clamp: both halves held to [0, 1], NaN -> 0
// (DX10_CLAMP).  Prints the results for a handful of operand pairs.  Build: hipcc --offload-arch=gfx950 -O2 pk_clamp.hip -o pk_clamp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2 *a, const f2 *b, f2 *o, f2 *o2)
{
    f2 x = a[threadIdx.x], y = b[threadIdx.x], r, q;
    asm volatile("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(q) : "v"(x), "v"(y));
    o[threadIdx.x] = r;
    o2[threadIdx.x] = q;
}
int main()
{
    const int n = 8;
    f2 ha[n] = {{0.25f, -0.5f}, {1.5f, 0.75f}, {-1e-20f, 1e-20f}, {NAN, 0.5f}, {0.3f, -INFINITY}, {-0.0f, 0.0f}, {1e-30f, 3e-39f}, {0.999999f, 1.0f}};
    f2 hb[n] = {{0.25f, 0.25f}, {0.0f, 0.5f}, {0.0f, 0.0f}, {0.0f, 0.25f}, {0.3f, 0.1f}, {0.0f, -0.0f}, {0.0f, 0.0f}, {1e-7f, 0.0f}};
    f2 *a, *b, *o, *o2, ho[n], ho2[n];
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&o, sizeof(ho)); hipMalloc(&o2, sizeof(ho2));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, a, b, o, o2);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(ho2, o2, sizeof(ho2), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("(%g, %g) + (%g, %g): plain (%g, %g)  clamp (%g, %g)\n", ha[i].x, ha[i].y, hb[i].x, hb[i].y, ho2[i].x, ho2[i].y, ho[i].x, ho[i].y);
    return 0;
}
