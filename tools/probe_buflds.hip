// probe: buffer_load_dword ... offen lds with out-of-range lanes: what lands in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* p, double* out, int nrec_bytes) {
  __shared__ double sh[256];
  for (int i = threadIdx.x; i < 256; i += 64) sh[i] = 7.0;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, nrec_bytes, 0x00020000);
  // lane l fetches dword (l ^ 5) of the source when l < 40, else out of range
  const unsigned l = threadIdx.x;
  unsigned voff = l < 40 ? (l ^ 5u) * 4u : 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sh, 4, voff, 0, 0, 0);
  // second instruction: +256 bytes in LDS, soffset 512 bytes into the source, lanes >= 48 beyond num_records by value
  unsigned voff2 = l * 4u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)((char*)sh + 256), 4, voff2, 512, 0, 0);
  // third: the scalar offset alone is beyond num_records
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)((char*)sh + 512), 4, voff2, 1024, 0, 0);
  // fourth: soffset == num_records
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)((char*)sh + 768), 4, voff2, nrec_bytes, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = sh[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = (float)i;
  float* d; double* o;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 256 * 8);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>((const double*)d, o, 512 + 48 * 4);    // num_records = 704 bytes
  std::vector<float> r(512);
  hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
  printf("first  (lanes 0..63 -> lds dwords 0..63):");
  for (int i = 0; i < 64; i++) printf(" %g", r[i]);
  printf("\nsecond (lds dwords 64..127):");
  for (int i = 64; i < 128; i++) printf(" %g", r[i]);
  printf("\nthird (soffset 1024 > num_records):");
  for (int i = 128; i < 192; i++) printf(" %g", r[i]);
  printf("\nfourth (soffset == num_records):");
  for (int i = 192; i < 256; i++) printf(" %g", r[i]);
  printf("\n");
  return 0;
}
