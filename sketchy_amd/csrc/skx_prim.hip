// skx_prim.hip -- the small device-wide primitives of the batch dictionary (sort / unique /
// exclusive scan), taken from rocPRIM (ROCm's own primitive library).  They touch tens of
// thousands of items per pass -- microseconds next to the multi-GB reference scan -- so they
// are library calls; the kernels of the hot path proper are hand-written in skx_kernels.hip.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "skx_kernels.hpp"

namespace skx {

size_t prim_scan_tmp_bytes(u32 n) {
    size_t b = 0;
    (void)rocprim::exclusive_scan(nullptr, b, (const u32*)nullptr, (u32*)nullptr, 0u, n, rocprim::plus<u32>());
    return b;
}
hipError_t prim_exclusive_scan_u32(hipStream_t st, void* tmp, size_t tmp_bytes, const u32* in, u32* out, u32 n) {
    if (n == 0) return hipSuccess;
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::plus<u32>(), st);
}

size_t prim_sort_tmp_bytes(u32 n) {
    size_t b = 0;
    (void)rocprim::radix_sort_keys(nullptr, b, (const u64*)nullptr, (u64*)nullptr, n);
    return b;
}
hipError_t prim_sort_u64(hipStream_t st, void* tmp, size_t tmp_bytes, const u64* in, u64* out, u32 n) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, n, 0, 64, st);
}

size_t prim_unique_tmp_bytes(u32 n) {
    size_t b = 0;
    (void)rocprim::unique(nullptr, b, (const u64*)nullptr, (u64*)nullptr, (u32*)nullptr, n,
                          rocprim::equal_to<u64>());
    return b;
}
hipError_t prim_unique_u64(hipStream_t st, void* tmp, size_t tmp_bytes, const u64* in, u64* out, u32* n_out, u32 n) {
    return rocprim::unique(tmp, tmp_bytes, in, out, n_out, n, rocprim::equal_to<u64>(), st);
}

size_t prim_segsort_tmp_bytes(u32 n, u32 n_seg) {
    size_t b = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, b, (const u64*)nullptr, (u64*)nullptr, n, n_seg, (const u32*)nullptr,
                                             (const u32*)nullptr);
    return b;
}
hipError_t prim_segsort_u64(hipStream_t st, void* tmp, size_t tmp_bytes, const u64* in, u64* out, u32 n, u32 n_seg,
                            const u32* seg_begin, const u32* seg_end) {
    if (n == 0 || n_seg == 0) return hipSuccess;
    return rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, in, out, n, n_seg, seg_begin, seg_end, 0, 64, st);
}

}  // namespace skx
