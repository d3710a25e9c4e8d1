// Probe: does the raw-buffer range check on gfx950 include soffset?  (DESIGN.md relies on it:
// the LDS-DMA staging puts the tile offset in soffset and expects rows past the end to read 0.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* out, int records_bytes, int soff) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records_bytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, threadIdx.x * 4, soff, 0) == 0 ? 0.0f
                       : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, threadIdx.x * 4, soff, 0));
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.0f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    // declare only the first 256 floats (1024 bytes) as records; read 64 floats at soffset 896 bytes:
    // lanes 0..31 -> byte 896..1020 (in range), lanes 32..63 -> 1024.. (out of range iff soffset counts)
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 1024, 896);
    std::vector<float> r(64);
    hipMemcpy(r.data(), o, 64 * 4, hipMemcpyDeviceToHost);
    printf("lane31=%g lane32=%g lane63=%g\n", r[31], r[32], r[63]);
    printf("%s\n", (r[31] == 1000.0f + 224 + 31 && r[32] == 0.0f && r[63] == 0.0f) ? "RANGE CHECK INCLUDES SOFFSET" : "RANGE CHECK IGNORES SOFFSET");
    return 0;
}
