// Work order of the similar-k-mer kernels (round 5): PfKmerArgs::order.
//
// A look-up into the offset table that misses the L2 costs one of the ~55 G memory-side requests per second the chip serves
// whatever the table size (profiles/r05_lookup_rate_probe.txt: 32 MB ... 1 GB all alike - the Infinity Cache does not help random
// 8 ... 32 byte requests; 2 MB, i.e. L2 hits, run at 250 G/s): the 1.0e9 look-ups of a 10 000-query batch WERE the 18 ms of that
// stage.  The k-mers of a window are a + n3 * b, a from the similar 3-mers of the window's first three residues, b from those of its
// last three: every window with the same last 3-mer reads the same stretches of the table (one stretch of n3 k-mers per similar b),
// and windows whose first 3-mers agree in their last two residues read nearly the same 32-byte blocks inside each stretch.  So
// the windows of a batch are processed SORTED by (last 3-mer, first 3-mer), and XCD x (its own 4 MB L2; workgroup b runs on XCD
// b % 8 - observed, not a contract, a wrong guess only costs speed) works through the x-th eighth of the sorted order
// (xcd_contiguous in pf_kernels.hip).  The lists go where they always went (list_base[gp]): only the order of the work changes.
//
// Shares: XCD x gets the x-th eighth of the sorted positions, i.e. a contiguous range of last-3-mer rows.  Dealing rows - or units of
// 20 neighbouring rows - round-robin evens the shares out (the similar k-mers of a window depend on its residues) but loses the
// overlap between the similar sets of neighbouring rows, which is worth more (profiles/r05_exp_pf_order.txt: contiguous 16.6 ms,
// units dealt 17.5, single rows dealt 18.9, no order 21.3).  Windows without k-mers (X, no window, profile positions) leave their
// wavefront at once; they are dealt over the shares by position and sort last in them.
// k = 7 (2 + 2 + 3 residues): last 3-mer = residues 4..6, "first" = the 2-mer of residues 2, 3.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

__global__ __launch_bounds__(256) void pf_order_keys_kernel(PfKmerArgs A, unsigned long long *keys) {
    const uint32_t gp = blockIdx.x * 256u + threadIdx.x;
    if (gp >= A.n_pos) return;
    uint32_t hi = 8191u, lo = 0u;
    if (A.q_thr[gp] >= 0 && !(A.q_kind && A.q_kind[gp])) {
        const uint8_t *q = A.q_res + gp;
        const uint32_t ka = A.kalph;
        if (A.k == 7) {
            hi = q[A.pat[4]] + ka * (q[A.pat[5]] + ka * q[A.pat[6]]);
            lo = q[A.pat[2]] + ka * q[A.pat[3]];
        } else if (A.k == 5) {
            hi = q[A.pat[2]] + ka * (q[A.pat[3]] + ka * q[A.pat[4]]);
            lo = q[A.pat[0]] + ka * q[A.pat[1]];
        } else {
            hi = q[A.pat[3]] + ka * (q[A.pat[4]] + ka * q[A.pat[5]]);
            lo = q[A.pat[0]] + ka * (q[A.pat[1]] + ka * q[A.pat[2]]);
        }
        hi = hi > 8190u ? 8190u : hi;      // (13 bits each: alphabets up to 20 letters fit exactly; larger ones only lose order, not work)
        lo = lo > 8191u ? 8191u : lo;
    }
    keys[gp] = ((unsigned long long)((hi << 13) | lo) << 32) | gp;
}

__global__ __launch_bounds__(256) void pf_order_extract_kernel(const unsigned long long *keys, uint32_t n, uint32_t *order) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) order[i] = (uint32_t)keys[i];
}

}  // namespace

// order[n_pos]; A needs q_res, q_thr, q_kind, pat, k, kalph.  Scratch comes from / goes back to `cache`.
hipError_t launch_pf_order(const PfKmerArgs &A, uint32_t *order, const std::shared_ptr<BlockCache> &cache, hipStream_t s) {
    if (A.n_pos == 0) return hipSuccess;
    DevBuf d_in, d_out, d_tmp;
    for (DevBuf *d : {&d_in, &d_out, &d_tmp}) d->bind(cache);
    hipError_t e = d_in.alloc((size_t)A.n_pos * 8);
    if (e == hipSuccess) e = d_out.alloc((size_t)A.n_pos * 8);
    if (e != hipSuccess) return e;
    size_t tmp_bytes = 0;
    e = rocprim::radix_sort_keys(nullptr, tmp_bytes, d_in.as<unsigned long long>(), d_out.as<unsigned long long>(), (size_t)A.n_pos, 32u, 61u, s);
    if (e == hipSuccess) e = d_tmp.alloc(std::max<size_t>(tmp_bytes, 16));
    if (e != hipSuccess) return e;
    const dim3 g((A.n_pos + 255) / 256), b(256);
    hipLaunchKernelGGL(pf_order_keys_kernel, g, b, 0, s, A, d_in.as<unsigned long long>());
    e = rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_in.as<unsigned long long>(), d_out.as<unsigned long long>(), (size_t)A.n_pos, 32u, 61u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pf_order_extract_kernel, g, b, 0, s, d_out.as<unsigned long long>(), A.n_pos, order);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // the scratch goes back to the shared cache at scope exit: nothing of this call may still be in flight then (a taker on
    // another stream would overwrite the sort's input); once per batch
    return hipStreamSynchronize(s);
}

}  // namespace mmgpu
