// What `buffer_load_dwordx4 ... lds` (LDS-DMA) writes on gfx950, for the bf16 GEMM staging of conv_gemm_bf16.hip:
//   (1) destination = wave-uniform LDS base (M0) + lane * 16, whatever the per-lane SOURCE offset is;
//   (2) a lane whose buffer offset is out of range (the 2^31 voffset the gather uses for padding taps / rows past the
//       end) writes ZEROS to its 16 bytes (it does not leave the old LDS bytes);
//   (3) an instruction offset moves the destination as well as the source (so it is not used).
// Build: hipcc -O3 --offload-arch=gfx950 -o tests/tools/glds_probe tests/tools/glds_probe.hip ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

__global__ void k(const unsigned* x, unsigned* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned smem[2 * 256 + 64];
  for (int i = threadIdx.x; i < 2 * 256 + 64; i += blockDim.x) smem[i] = 0xABABABABu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)nbytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // source: lane l reads chunk (63 - l) (a permutation); every third lane is out of range
  unsigned voff = (unsigned)(63 - lane) * 16u + (unsigned)wave * 1024u;
  if (lane % 3 == 2) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void*)(smem + wave * 256), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 256 + 64; i += blockDim.x) out[i] = smem[i];
}

int main() {
  const int n = 2 * 256;
  std::vector<unsigned> h(n), o(n + 64);
  for (int i = 0; i < n; ++i) h[i] = 0x10000u + i;
  unsigned *dx, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, (n + 64) * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, dx, dout, (unsigned)(n * 4));
  hipMemcpy(o.data(), dout, (n + 64) * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const unsigned got = o[w * 256 + l * 4 + j];
        const unsigned want = (l % 3 == 2) ? 0u : 0x10000u + w * 256 + (63 - l) * 4 + j;
        if (got != want) { if (bad < 8) printf("wave %d lane %d dword %d: got %08x want %08x\n", w, l, j, got, want); ++bad; }
      }
  for (int i = 0; i < 64; ++i) if (o[n + i] != 0xABABABABu) { printf("tail %d overwritten: %08x\n", i, o[n + i]); ++bad; }
  printf("glds_probe: %s (%d mismatches): dest = M0 base + lane*16, out-of-range lanes write zeros\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
