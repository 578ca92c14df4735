// Persisted device layout (mmgpu_db_save / mmgpu_db_load, pf_api.hip): what protects the kernels
// from a damaged or foreign file.  A section's checksum is taken on the device where the section lies (at save time from the
// resident arrays, at load time from what the copy engine delivered) and the layout is checked for the properties the kernels
// rely on - a target's residues inside the residue block, offsets non-decreasing and ending at the entry count, entry ids below
// the target count - before the context adopts it.  HBM-bound, a few milliseconds for the 3 GB of a 1 M-target database.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {      // (splitmix64's finaliser)
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// sum over the 8-byte words of mix(word + golden * (index + 1)): order-independent to compute, position-dependent in value
__global__ __launch_bounds__(256) void db_checksum_kernel(const uint64_t *words, uint64_t n_words, const uint8_t *tail, uint32_t n_tail,
                                                          unsigned long long *out) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256u)
        acc += mix64(words[i] + 0x9E3779B97F4A7C15ull * (i + 1));
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_tail) {
        uint64_t w = 0;
        for (uint32_t k = 0; k < n_tail; k++) w |= (uint64_t)tail[k] << (8 * k);
        acc += mix64(w + 0x9E3779B97F4A7C15ull * (n_words + 1));
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor((unsigned long long)acc, d);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// bad[0] |= 1: a target reaches beyond the residue block / is longer than max_len; 2: offsets decrease or pass n_entries;
// 4: the table does not end at n_entries; 8: an entry names a target that is not there
__global__ __launch_bounds__(256) void db_validate_kernel(const uint32_t *off4, const uint32_t *len, uint32_t n, uint64_t res_bytes, uint32_t max_len,
                                                          const uint32_t *offsets, uint64_t table, const uint64_t *entries, uint64_t n_entries,
                                                          uint32_t *bad) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    uint32_t b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += stride)
        if (len[i] > max_len || (uint64_t)off4[i] * 4u + len[i] > res_bytes) b |= 1u;
    if (offsets) {
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < table; i += stride)
            if (offsets[i] > offsets[i + 1] || offsets[i + 1] > n_entries) b |= 2u;
        if (blockIdx.x == 0 && threadIdx.x == 0 && offsets[table] != n_entries) b |= 4u;
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_entries; i += stride)
            if ((uint32_t)entries[i] >= n) b |= 8u;
    }
    if (b) atomicOr(bad, b);
}

}  // namespace

// *sum (host) = checksum of `bytes` bytes at `dev`; scratch = 8 bytes of device memory; synchronises the stream
hipError_t db_section_checksum(const void *dev, size_t bytes, unsigned long long *scratch, uint64_t *sum, hipStream_t s) {
    hipError_t e = hipMemsetAsync(scratch, 0, 8, s);
    if (e != hipSuccess) return e;
    const uint64_t n_words = bytes / 8;
    const unsigned grid = (unsigned)std::min<uint64_t>(4096, std::max<uint64_t>(1, (n_words + 255) / 256));
    hipLaunchKernelGGL(db_checksum_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint64_t *>(dev), n_words,
                       static_cast<const uint8_t *>(dev) + n_words * 8, (uint32_t)(bytes - n_words * 8), scratch);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    unsigned long long h = 0;
    if ((e = hipMemcpyAsync(&h, scratch, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    *sum = h;
    return hipSuccess;
}

// *bad (host) = the flags of db_validate_kernel; offsets may be null (targets only); scratch = 4 bytes of device memory
hipError_t db_validate_layout(const uint32_t *off4, const uint32_t *len, uint32_t n, uint64_t res_bytes, uint32_t max_len, const uint32_t *offsets,
                              uint64_t table, const uint64_t *entries, uint64_t n_entries, uint32_t *scratch, uint32_t *bad, hipStream_t s) {
    hipError_t e = hipMemsetAsync(scratch, 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(db_validate_kernel, dim3(4096), dim3(256), 0, s, off4, len, n, res_bytes, max_len, offsets, table, entries, n_entries, scratch);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(bad, scratch, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

}  // namespace mmgpu
