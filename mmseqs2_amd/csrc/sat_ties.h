// QueryMatcher.cpp:147-177 for one nucleotide query whose saturated elements tie (pf_keepmax_nucl_kernel): the reference sorts its
// saturated elements by target id with std::sort and gives a target the diagonal of the FIRST of its elements, in the order the
// sort left, that reaches the best exact score.  Up to 16 elements libstdc++ sorts by insertion (stable); beyond, the order of
// equal ids belongs to its introsort.  The same std::sort (this library and the reference are built against the same libstdc++)
// over the same elements in the same initial order - the reference's array order: cache bin, then arrival - gives it.
// Plain C++, no device code: mmgpu_pf_fetch uses it, tests/test_nucl_prefilter.py compiles it for the host and checks it
// against the real reference.
#ifndef MMGPU_SAT_TIES_H
#define MMGPU_SAT_TIES_H

#include <stdint.h>

#include <algorithm>
#include <vector>

namespace mmgpu {

// E has the members id, arr (arrival index), score (exact ungapped score), diag.  After the call the first element of every id
// group of `el` carries the diagonal the reference keeps for that target.
template <typename E>
inline void resolve_saturated_ties(std::vector<E> &el, uint32_t refmask) {
    std::sort(el.begin(), el.end(), [refmask](const E &x, const E &y) {      // the order before the reference's sort (keys are unique)
        const uint64_t kx = ((uint64_t)(x.id & refmask) << 32) | x.arr, ky = ((uint64_t)(y.id & refmask) << 32) | y.arr;
        return kx < ky;
    });
    std::sort(el.begin(), el.end(), [](const E &x, const E &y) { return x.id < y.id; });      // CounterResult::sortById
    uint32_t prev = 0xFFFFFFFFu;
    size_t first = 0;
    uint64_t best = 0;
    for (size_t i = 0; i < el.size(); i++) {      // :158-171
        if (prev == el[i].id) {
            if ((uint64_t)el[i].score > best) {
                best = el[i].score;
                el[first].diag = el[i].diag;
            }
        } else {
            best = (i + 1 < el.size() && el[i + 1].id == el[i].id) ? el[i].score : 0;
            first = i;
        }
        prev = el[i].id;
    }
}

}  // namespace mmgpu

#endif
