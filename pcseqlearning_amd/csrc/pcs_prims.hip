// Device-wide sort / scan primitives. This is the only translation unit that includes rocPRIM
// (vendor primitive library, used the way a GEMM would use hipBLASLt); every domain kernel
// (voxel keys, cell walk, rulebook, conv, ICP) is hand-written in the other .hip files.
#include "pcs_common.hpp"
#include <rocprim/rocprim.hpp>

namespace pcs {

size_t sort_pairs_u64_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                            (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64, 0);
  return bytes;
}

int sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout,
                   const int32_t* vin, int32_t* vout, size_t n, int begin_bit, int end_bit,
                   hipStream_t s) {
  if (n == 0) return PCS_OK;
  hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n,
                                           (unsigned)begin_bit, (unsigned)end_bit, s);
  return e == hipSuccess ? PCS_OK : PCS_ERR_HIP;
}

size_t sort_pairs_u32_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                            (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 32, 0);
  return bytes;
}

int sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout,
                   const int32_t* vin, int32_t* vout, size_t n, int begin_bit, int end_bit,
                   hipStream_t s) {
  if (n == 0) return PCS_OK;
  hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n,
                                           (unsigned)begin_bit, (unsigned)end_bit, s);
  return e == hipSuccess ? PCS_OK : PCS_ERR_HIP;
}

size_t scan_i32_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0,
                          n, rocprim::plus<int32_t>(), 0);
  size_t b2 = 0;
  (void)rocprim::inclusive_scan(nullptr, b2, (const int32_t*)nullptr, (int32_t*)nullptr, n,
                          rocprim::plus<int32_t>(), 0);
  return bytes > b2 ? bytes : b2;
}

int exclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n,
                       hipStream_t s) {
  if (n == 0) return PCS_OK;
  hipError_t e =
      rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), s);
  return e == hipSuccess ? PCS_OK : PCS_ERR_HIP;
}

int inclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n,
                       hipStream_t s) {
  if (n == 0) return PCS_OK;
  hipError_t e = rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<int32_t>(), s);
  return e == hipSuccess ? PCS_OK : PCS_ERR_HIP;
}

size_t scan_i64_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                          n, rocprim::plus<int64_t>(), 0);
  return bytes;
}

int exclusive_scan_i64(void* tmp, size_t tmp_bytes, const int64_t* in, int64_t* out, size_t n,
                       hipStream_t s) {
  if (n == 0) return PCS_OK;
  hipError_t e =
      rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), s);
  return e == hipSuccess ? PCS_OK : PCS_ERR_HIP;
}

}  // namespace pcs

namespace pcs {
void* thread_pinned_page() {
  static thread_local void* page = nullptr;
  static thread_local bool tried = false;
  if (!tried) {
    tried = true;
    if (hipHostMalloc(&page, kPinnedPageBytes, hipHostMallocDefault) != hipSuccess) page = nullptr;
  }
  return page;
}
}  // namespace pcs
