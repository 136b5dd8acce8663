// Micro-benchmark (round 5): how many 256-thread blocks of a kernel with N bytes of static LDS does the runtime place on one gfx950 CU?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor; the LDS is 160 KB = 163840 B)   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_occ.hip -o /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> __global__ __launch_bounds__(256) void k(float *o)
{
    __shared__ float s[N / 4];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    o[threadIdx.x] = s[(threadIdx.x * 7) % (N / 4)];
}
template <int N> void q()
{
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<N>, 256, 0);
    printf("static LDS %6d B: %d blocks per CU (%s)\n", N, nb, hipGetErrorString(e));
}
int main()
{
    q<32768>(); q<38192>(); q<40960>(); q<41984>(); q<53248>(); q<54528>(); q<65536>(); q<76800>(); q<77664>(); q<78528>(); q<79360>(); q<80640>(); q<81920>(); q<83200>();
    return 0;
}
