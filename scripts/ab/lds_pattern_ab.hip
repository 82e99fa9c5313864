// Which lanes of a wave64 does the LDS serve together?  Times ds_read_b128 / ds_read_b64 / ds_read_b32 with controlled bank patterns.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ab/lds_pattern_ab.hip -o build/lds_pattern_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int W>   // W = dwords per read (1, 2, 4)
__global__ void k(const int* __restrict__ off, unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int o = off[threadIdx.x];
    float acc = 0.f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int a = (o + it * 0) & 8191;
        if (W == 4) { const float4 v = *reinterpret_cast<const float4*>(s + a); acc += v.x + v.y + v.z + v.w; }
        if (W == 2) { const float2 v = *reinterpret_cast<const float2*>(s + a); acc += v.x + v.y; }
        if (W == 1) { acc += s[a]; }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}

int main() {
    int* d_off; unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 8); hipMalloc(&d_sink, 64 * 4);
    const int iters = 20000;
    auto run = [&](const char* name, int W, std::vector<int> off) {
        hipMemcpy(d_off, off.data(), 64 * 4, hipMemcpyHostToDevice);
        unsigned long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (W == 4) k<4><<<1, 64>>>(d_off, d_out, d_sink, iters);
            if (W == 2) k<2><<<1, 64>>>(d_off, d_out, d_sink, iters);
            if (W == 1) k<1><<<1, 64>>>(d_off, d_out, d_sink, iters);
            hipDeviceSynchronize();
            hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
        }
        printf("%-72s W=%d  %.1f cycles/read\n", name, W, (double)c / iters);
    };
    std::vector<int> o(64);
    // ---- b128: bank group g = (dword/4) % 8 ----
    for (int l = 0; l < 64; ++l) o[l] = l * 4;                         run("b128 linear (conflict-free)", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = 0;                             run("b128 all same address (broadcast)", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = (l % 8) * 4 + (l / 8) * 32 * 4; run("b128 octet of consecutive lanes distinct groups, octets collide", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = (l / 8) * 4 + (l % 8) * 32 * 4; run("b128 lanes l, l+8, .. distinct groups; consecutive lanes same group", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = ((l / 2) % 8) * 4 + (l % 2) * 32 * 4 + (l / 16) * 64 * 4; run("b128 pairs of consecutive lanes share a group (2-way in octet)", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = (l % 4) * 4 + (l / 4) * 32 * 4; run("b128 only 4 groups used by consecutive lanes (2-way per octet)", 4, o);
    for (int l = 0; l < 64; ++l) o[l] = (l / 16) * 4 + (l % 16) * 32 * 4; run("b128 16 consecutive lanes same group (worst)", 4, o);
    srand(1);
    for (int l = 0; l < 64; ++l) o[l] = (rand() % 55) * 12;            run("b128 random joints, stride 12 (the skinning pattern)", 4, o);
    // ---- b64: pair position = (dword/2) % 16 ----
    for (int l = 0; l < 64; ++l) o[l] = l * 2;                         run("b64 linear", 2, o);
    for (int l = 0; l < 64; ++l) o[l] = (l % 16) * 2 + (l / 16) * 32 * 2; run("b64 16 consecutive lanes distinct, quarters collide", 2, o);
    for (int l = 0; l < 64; ++l) o[l] = (l / 4) * 2 + (l % 4) * 32 * 2; run("b64 4 consecutive lanes same position", 2, o);
    for (int l = 0; l < 64; ++l) o[l] = (l % 32) * 2 * 0 + ((l % 32) / 2) * 2 + (l % 2) * 64 + (l / 32) * 128; run("b64 2 consecutive lanes same position", 2, o);
    for (int l = 0; l < 64; ++l) o[l] = (rand() % 1024) * 6;            run("b64 random vertices, stride 6 (the entry-walk pattern)", 2, o);
    // ---- b32 ----
    for (int l = 0; l < 64; ++l) o[l] = l;                             run("b32 linear", 1, o);
    for (int l = 0; l < 64; ++l) o[l] = (l % 32) + (l / 32) * 64;      run("b32 halves collide bankwise (l, l+32 same bank)", 1, o);
    for (int l = 0; l < 64; ++l) o[l] = (l / 2) + (l % 2) * 32;        run("b32 consecutive lanes same bank", 1, o);
    for (int l = 0; l < 64; ++l) o[l] = (rand() % 55);                 run("b32 random joints, transposed table (index = joint)", 1, o);
    return 0;
}
