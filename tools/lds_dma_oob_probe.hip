// Probe (run on the GPU box): what does `buffer_load_dwordx4 ... lds` write for lanes whose offset is outside the buffer
// resource's num_records?  Expected (raw buffer, stride 0): zeros.  Prints the LDS image of one 1-KiB piece.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_dma_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, (int)nbytes, 0x00020000);
    // even lanes: in range (lane * 16); odd lanes: far out of range
    const unsigned off = (lane & 1) ? 0x7ffffff0u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    // second piece, immediate offset 1024 on both sides: global +1024 (out of range for a 1-KiB buffer -> zeros), LDS +1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (unsigned)lane * 16u, 0, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main()
{
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    unsigned *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 2048);
    hipMemset(d, 0x55, 4096);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 1024, o);
    std::vector<unsigned> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int ok_even = 0, zero_odd = 0, zero_second = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            const unsigned v = r[l * 4 + k];
            if (!(l & 1)) ok_even += v == 0x1000u + l * 4 + k;
            else zero_odd += v == 0;
            zero_second += r[256 + l * 4 + k] == 0;
        }
    printf("in-range lanes correct: %d/128, out-of-range lanes zero: %d/128 (first words: %08x %08x), second piece (imm 1024, all out of range) zero: %d/256 (first %08x)\n",
           ok_even, zero_odd, r[4], r[5], zero_second, r[256]);
    return 0;
}
