// Developer experiment: where do workgroups land?  Records HW_ID / XCC_ID / start time of wave 0 of every workgroup
// for a launch shaped like k_conv (256 threads, ~68 KB LDS => 2 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(uint32_t* out, int spin) {
  __shared__ float pad[17000];
  uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  uint64_t t0 = __builtin_readcyclecounter();
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
  pad[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x % 64 == 0) {
    int w = threadIdx.x / 64;
    uint32_t* o = out + (blockIdx.x * 4 + w) * 4;
    o[0] = hw; o[1] = xcc; o[2] = (uint32_t)(t0 >> 6); o[3] = (uint32_t)pad[(threadIdx.x + 1) % 256];
  }
}
int main() {
  const int nb = 2048;
  uint32_t* d; hipMalloc(&d, nb * 16 * sizeof(uint32_t));
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(nb * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  printf("block wave | xcc se sh cu simd slot | t0/64\n");
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 4; ++w) {
      uint32_t hw = h[(b * 4 + w) * 4], x = h[(b * 4 + w) * 4 + 1], t = h[(b * 4 + w) * 4 + 2];
      if (b < 40 || (b >= 512 && b < 530) || b % 257 == 0)
        printf("%5d %d | %u %u %u %2u %u %u | %u\n", b, w, x & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, t);
    }
  // co-residency: for first-round blocks, group by (xcc,se,sh,cu)
  int same_par = 0, diff_par = 0, slots[16] = {0};
  for (int b = 0; b < 512; ++b)
    for (int c = b + 1; c < 512; ++c) {
      uint32_t hb = h[b * 16], hc = h[c * 16];
      if ((h[b * 16 + 1] & 15) == (h[c * 16 + 1] & 15) && ((hb >> 8) & 0xff) == ((hc >> 8) & 0xff)) {
        if ((b & 1) == (c & 1)) same_par++; else diff_par++;
      }
    }
  for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) slots[h[(b * 4 + w) * 4] & 15]++;
  printf("first-round co-resident pairs: same blockIdx parity %d, different %d\n", same_par, diff_par);
  printf("wave slot histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", slots[i]); printf("\n");
  return 0;
}
