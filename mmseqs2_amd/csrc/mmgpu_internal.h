// Internal declarations shared by the HIP translation units of libmmgpu (not part of the C-ABI).
#ifndef MMGPU_INTERNAL_H
#define MMGPU_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmgpu.h"

namespace mmgpu {

// Target database resident in HBM.  Every sequence starts on a 4-byte boundary so the group-head lane can
// fetch four residues per global_load_dword; `off4` is the start in dwords.
struct DeviceDb {
    uint8_t *res = nullptr;      // residues, 4-byte aligned starts, tail slack of max_len + 64 bytes
    uint32_t *off4 = nullptr;    // [n] start of sequence i in units of 4 bytes
    uint32_t *len = nullptr;     // [n]
    uint32_t n = 0;
    uint32_t max_len = 0;
    uint64_t total_residues = 0;
    int alphabet = 0;
};

// One workgroup's share of a batch: hits [hit_begin, hit_end) of one query (already sorted by target length).
struct SwJob {
    uint32_t query;
    uint32_t hit_begin;
    uint32_t hit_end;
    uint32_t pad;
};

// Everything a Smith-Waterman launch needs, device pointers only.
struct SwLaunch {
    const SwJob *jobs;
    uint32_t n_jobs;
    // queries of the batch
    const uint8_t *q_res;     // concatenated numeric residues
    const int8_t *q_cb;       // concatenated rounded composition bias (zeros when absent)
    const uint32_t *q_off;    // [nq+1]
    const int32_t *q_bias;    // [nq] ssw_init bias (StripedSmithWaterman.cpp:1397-1406), decides `word`
    const int32_t *q_minstart; // [nq] reverse scan only for pairs with score >= this
    // targets
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const uint32_t *t_len;
    // hits (sorted order) and where each one's result goes
    const uint32_t *hit_target;
    const uint32_t *hit_out;  // index into out[]
    mmgpu_sw_hit *out;
    // reverse pass only: per-hit forward result lives in out[hit_out[h]]
    // scoring
    const int8_t *mat;        // alphabet*alphabet
    int alphabet;
    int gap_open, gap_extend;
    // multi-tile scratch (H/F boundary rows), [n_jobs][4 waves][4 groups][scratch_cols] x uint2
    uint2 *scratch;
    uint32_t scratch_cols;
};

// rows_per_lane in {8,16,24,32}: a 16-lane group covers 16*rows_per_lane query rows per tile.
// reverse = false: forward score/end scan; true: start-position scan over the reversed prefixes.
hipError_t launch_sw(const SwLaunch &L, int rows_per_lane, bool multi_tile, bool reverse, hipStream_t stream);
size_t sw_lds_bytes(int rows_per_lane, int alphabet);

}  // namespace mmgpu
#endif
