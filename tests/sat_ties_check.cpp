// Host build of mmseqs2_amd/csrc/sat_ties.h for tests/test_nucl_prefilter.py: elements in, the diagonal kept per target out.
#include <stdint.h>

#include <vector>

#include "sat_ties.h"

struct Elem {
    uint32_t id, arr, score;
    uint16_t diag;
};

extern "C" int sat_ties_resolve(const uint32_t *id, const uint32_t *arr, const uint32_t *score, const uint16_t *diag, uint32_t n, uint32_t refmask,
                                uint32_t *out_id, uint16_t *out_diag) {
    std::vector<Elem> el(n);
    for (uint32_t i = 0; i < n; i++) {
        el[i].id = id[i];
        el[i].arr = arr[i];
        el[i].score = score[i];
        el[i].diag = diag[i];
    }
    mmgpu::resolve_saturated_ties(el, refmask);
    int groups = 0;
    uint32_t prev = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) {
        if (el[i].id == prev) continue;
        prev = el[i].id;
        out_id[groups] = el[i].id;
        out_diag[groups] = el[i].diag;
        groups++;
    }
    return groups;
}
