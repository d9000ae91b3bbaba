// Lab: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS (raw buffer, stride 0)?
// hipcc --offload-arch=gfx950 -O3 tools/buf_oob_lab.hip -o build/lab/buf_oob_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr;
__global__ void k(const char* a, unsigned nrec, const unsigned* voffs, int soff, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[1024];
    for (int i = threadIdx.x; i < 256; i += 64) ((float*)smem)[i] = -7.f;      // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)smem, 16, voffs[threadIdx.x], soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = 1.f + i;
    char* a; float* out; unsigned* vo;
    hipMalloc(&a, 16384); hipMemcpy(a, h.data(), 16384, hipMemcpyHostToDevice);
    hipMalloc(&out, 1024); hipMalloc(&vo, 256);
    auto run = [&](const char* name, unsigned nrec, int soff, unsigned oob) {
        std::vector<unsigned> v(64);
        for (int l = 0; l < 64; l++) v[l] = (l % 4 == 3) ? oob : l * 16;
        hipMemcpy(vo, v.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, nrec, vo, soff, out);
        std::vector<float> o(256);
        hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
        int bad = 0, zero = 0, poison = 0;
        for (int l = 0; l < 64; l++)
            for (int e = 0; e < 4; e++) {
                const float got = o[l * 4 + e];
                if (l % 4 == 3) { zero += got == 0.f; poison += got == -7.f; }
                else bad += got != h[(l * 16 + soff) / 4 + e];
            }
        printf("%-40s in-range mismatches %d; OOB lanes: zero %d/64, untouched %d/64\n", name, bad, zero, poison);
    };
    run("nrec=16384 oob=0xffffffff soff=0", 16384, 0, 0xffffffffu);
    run("nrec=16384 oob=0x80000000 soff=0", 16384, 0, 0x80000000u);
    run("nrec=0xffffffff oob=0xffffffff soff=0", 0xffffffffu, 0, 0xffffffffu);
    run("nrec=0xffffffff oob=0xffffffff soff=128", 0xffffffffu, 128, 0xffffffffu);
    run("nrec=0xfffffff0 oob=0xfffffff0 soff=128", 0xfffffff0u, 128, 0xfffffff0u);
    run("nrec=0x80000000 oob=0x80000000 soff=128", 0x80000000u, 128, 0x80000000u);
    return 0;
}
