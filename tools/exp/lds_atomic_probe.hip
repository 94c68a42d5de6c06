// What does SQ_LDS_BANK_CONFLICT count for LDS atomics?  (tools/exp: a measurement program, never shipped.)
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_atomic_probe tools/exp/lds_atomic_probe.hip
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d out -o p -- build/lds_atomic_probe
// Each kernel issues 256 ds_add_u32 per wave with a fixed lane -> bin pattern:
//   p0: bin = lane (all distinct, consecutive)          p1: bin = lane / 2        p2: lane / 4      p3: lane / 8
//   p4: lane / 16                                        p5: lane / 64 (one bin)   p6: bin = lane * 32 (one bank, distinct addresses)
//   p7: the rotated pattern: even lanes lane / 16, odd lanes 512 + lane / 16      p8: b64 reads consecutive   p9: b128 reads consecutive
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int P>
__global__ void probe(unsigned int *out) {
    __shared__ __attribute__((aligned(16))) unsigned int h[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int bin = P == 0 ? lane : P == 1 ? lane / 2 : P == 2 ? lane / 4 : P == 3 ? lane / 8 : P == 4 ? lane / 16 : P == 5 ? 0 : P == 6 ? lane * 32
              : (lane & 1) ? 512 + lane / 16 : lane / 16;
    unsigned int acc = 0;
    if (P <= 7) {
        for (int k = 0; k < 256; ++k) atomicAdd(&h[(bin + k * 7) & 4095], 1u);
    } else if (P == 8) {
        const double *d = reinterpret_cast<const double *>(h);
        double s = 0;
        for (int k = 0; k < 256; ++k) s += d[(lane + k * 64) & 2047];
        acc = (unsigned int)s;
    } else {
        const uint4 *d = reinterpret_cast<const uint4 *>(h);
        for (int k = 0; k < 256; ++k) { const uint4 v = d[(lane + k * 64) & 1023]; acc += v.x + v.y + v.z + v.w; }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = h[threadIdx.x] + acc;
}
int main() {
    unsigned int *d;
    hipMalloc(&d, 256 * 64 * 4);
#define RUN(P) hipLaunchKernelGGL(probe<P>, dim3(256), dim3(64), 0, 0, d); hipDeviceSynchronize();
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    printf("done\n");
    return 0;
}
