// Banded traceback (CIGAR) for gfx950: SmithWaterman::banded_sw + its traceback
// (src/alignment/StripedSmithWaterman.cpp:1478-1693, SEQ_SEQ branch) and the identity count of computerBacktrace
// (:1280-1308), bit for bit.
//
// The reference runs a scalar int32 Gotoh over the sub-rectangle [q_start..q_end] x [t_start..t_end] inside a band
// of half-width |tlen - qlen| + 1 that is doubled until the banded maximum reaches the Smith-Waterman score, storing
// three direction bytes per cell, then walks back from the bottom-right corner.  Its tie rules (E/F prefer the
// extension on a tie :1551-1557, H prefers the diagonal :1575, E over F only if strictly greater :1576) and its
// band-edge handling (one zeroed cell per row, :1528) decide the CIGAR, so the kernel keeps the reference's own
// arrays and index arithmetic:
//   * one LANE per alignment (the host sorts the jobs by size so the 64 lanes of a wavefront run similar loops);
//     the DP is a short-band recurrence with a serial F chain - there is nothing to vectorise inside one alignment
//     and there are up to 300 x n_queries independent ones;
//   * the three band rows (H previous, E, H current; H rows ping-pong instead of being copied) live in LDS for bands up
//     to 32 (first tier: nearly every job), lane-interleaved (word k of lane l at k * 64 + l: no bank conflicts), in
//     a per-wavefront scratch area in HBM with the same interleave for wider bands; the direction bits always go to
//     that scratch (lanes that sit at the same band offset - the loops are driven by the band index - coalesce);
//   * directions are packed to 4 bits per cell (E bit, F bit, 2-bit H source) instead of 3 bytes: 0.5 B/cell written.
// Work per alignment ~ qlen * (2 * band + 1) cells, only for the pairs the host asks a backtrace for
// (Matcher::SCORE_COV_SEQID / -a), so this kernel is bounded by per-lane latency, not by HBM or VALU peak.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

// LDS_ROWS: the three band rows live in LDS (bands up to BT_LDS_BAND wide - nearly every alignment of a hit list: the
// band is |tlen - qlen| + 1); every cell reads / writes them in a dependent chain, which through HBM-backed scratch costs a
// memory round trip per access.  Wider bands keep the rows in the lane-interleaved global scratch (the second and
// third tier).  The direction bits are write-only during the DP and go to global scratch in both forms.
constexpr int BT_LDS_WIDTH = 67;     // 2 * band + 3 words per row, band <= 32

template <bool LDS_ROWS>
__global__ __launch_bounds__(64) void sw_traceback_kernel(BtLaunch L) {
    __shared__ int8_t smat[32 * 32];
    __shared__ uint32_t srows[LDS_ROWS ? 3 * BT_LDS_WIDTH * 64 : 1];
    const int lane = (int)threadIdx.x;
    for (int k = lane; k < 32 * 32; k += 64) smat[k] = k < L.alphabet * L.alphabet ? L.mat[k] : (int8_t)0;
    __syncthreads();
    const uint32_t jidx = blockIdx.x * 64u + (uint32_t)lane;
    if (jidx >= L.n_jobs) return;
    const BtJob J = L.jobs[jidx];
    const uint32_t ji = J.slot;
    uint32_t *W = L.scratch + (size_t)blockIdx.x * L.words_per_lane * 64u;
#define DIRW(idx) W[(size_t)(idx) * 64u + (uint32_t)lane]
#define AT(idx) (LDS_ROWS ? srows[(uint32_t)(idx) * 64u + (uint32_t)lane] : W[(size_t)(idx) * 64u + (uint32_t)lane])
    const int ql = J.q_end - J.q_start + 1, tl = J.t_end - J.t_start + 1;
    const uint8_t *q = L.q_res + L.q_off[J.query] + J.q_start;
    const int8_t *cb = L.q_cb + L.q_off[J.query] + J.q_start;
    const uint8_t *t = L.t_res + (size_t)L.t_off4[J.target] * 4 + J.t_start;
    const int go = L.gap_open, ge = L.gap_extend, alph = L.alphabet;
    // profile query: score of (row i, letter t) = prof[t * full query length + q_start + i], no bias (:1565-1567)
    const int8_t *prof = nullptr;
    int qfull = 0;
    if (L.q_prof_off && L.q_prof_off[J.query] != 0xFFFFFFFFu) {
        qfull = (int)(L.q_off[J.query + 1] - L.q_off[J.query]);
        prof = L.q_prof + L.q_prof_off[J.query] + J.q_start;
    }
    const uint32_t HC = LDS_ROWS ? (uint32_t)BT_LDS_WIDTH : L.band_cap;   // words per band row
    const uint32_t DIR0 = LDS_ROWS ? 0u : 3u * HC;  // direction words start here (the LDS form keeps no rows in W)
    const uint64_t DIRCAP = (uint64_t)L.words_per_lane - DIR0;
    mmgpu_sw_bt info;
    info.bt_off = J.bt_off;
    info.bt_len = 0;
    info.ident = 0;
    info.status = MMGPU_BT_OK;
    info.reserved = 0;

    int bw = (tl > ql ? tl - ql : ql - tl) + 1;
    int maxv = 0;
    uint32_t row_words = 0;
    bool fail = false;
    do {
        const int64_t width = (int64_t)bw * 2 + 3, width_d = (int64_t)bw * 2 + 1;
        row_words = (uint32_t)((width_d + 7) / 8);
        if ((uint64_t)width > HC || (uint64_t)row_words * (uint64_t)ql > DIRCAP) {
            fail = true;
            break;
        }
        uint32_t hb = 0, hc = 2u * HC;              // H previous / H current (ping-pong), E at HC
        const uint32_t eb = HC;
        for (int64_t k = 0; k < width; k++) {
            AT(hb + k) = 0;
            AT(eb + k) = 0;
            AT(hc + k) = 0;
        }
        maxv = 0;
        for (int i = 0; i < ql; i++) {
            const int beg = max(0, i - bw), end = min(tl - 1, i + bw);
            const int edge = (int)min((int64_t)end + 1, width - 1);
            AT(hb) = 0;
            AT(eb) = 0;
            AT(hb + edge) = 0;
            AT(eb + edge) = 0;
            AT(hc) = 0;
            const int sh_i = max(0, i - bw), sh_p = max(0, i - 1 - bw);   // band shift of this / the previous row
            const int qi = (int)q[i] * alph;
            const int cbi = (int)cb[i];
            int f = 0, hleft = 0;                   // h_c[b]: H of the cell to the left (h_c[0] = 0)
            uint32_t word = 0;
            const uint32_t dbase = DIR0 + (uint32_t)i * row_words;
            uint8_t t_next = t[beg];                // the letter of the next cell is requested one cell ahead
            for (int j = beg; j <= end; j++) {
                const uint8_t t_cur = t_next;
                if (j < end) t_next = t[j + 1];
                const int u = j - sh_i + 1, e = j - sh_p + 1, d = e - 1;
                const int x = j - sh_i;
                int temp1 = i == 0 ? -go : (int)AT(hb + e) - go;
                int temp2 = i == 0 ? -ge : (int)AT(eb + e) - ge;
                const int ev = temp1 > temp2 ? temp1 : temp2;
                const uint32_t de = temp1 > temp2 ? 1u : 0u;          // 3 : 2
                AT(eb + u) = (uint32_t)ev;
                temp1 = hleft - go;
                temp2 = f - ge;
                f = temp1 > temp2 ? temp1 : temp2;
                const uint32_t df = temp1 > temp2 ? 1u : 0u;          // 5 : 4
                const int f1 = f > 0 ? f : 0;
                const int e1 = ev > 0 ? ev : 0;
                temp1 = e1 > f1 ? e1 : f1;
                temp2 = (int)AT(hb + d) + (prof ? (int)prof[(int)t_cur * qfull + i] : (int)smat[qi + (int)t_cur] + cbi);
                const int h = temp1 > temp2 ? temp1 : temp2;
                AT(hc + u) = (uint32_t)h;
                hleft = h;
                maxv = h > maxv ? h : maxv;
                const uint32_t hsel = temp1 <= temp2 ? 0u : (e1 > f1 ? 1u : 2u);
                word |= (de | (df << 1) | (hsel << 2)) << ((x & 7) * 4);
                if ((x & 7) == 7 || j == end) {
                    DIRW(dbase + (uint32_t)(x >> 3)) = word;
                    word = 0;
                }
            }
            const uint32_t tmp = hb;                // h_b <- h_c (:1584), as a swap
            hb = hc;
            hc = tmp;
        }
        bw *= 2;
    } while (maxv < J.score);
    if (fail) {
        info.status = MMGPU_BT_TOO_LARGE;
        L.info[ji] = info;
        return;
    }
    bw /= 2;

    // traceback from the bottom-right corner in state H (:1590-1651); ops are produced last to first
    char *out = L.bt + J.bt_off;
    const uint32_t cap = (uint32_t)(ql + tl + 1);
    uint32_t n = 0, ident = 0;
    int i = ql - 1, j = tl - 1, state = 2;
    bool ok = true;
    while (i > 0 || j > 0) {
        const int sh = max(0, i - bw);
        const int x = j - sh;
        if (i < 0 || j < 0 || x < 0 || j > min(tl - 1, i + bw) || n + 1 >= cap) {
            ok = false;   // the walk left the band: the reference would read unrelated direction bytes here
            break;
        }
        const uint32_t w = DIRW(DIR0 + (uint32_t)i * row_words + (uint32_t)(x >> 3));
        const uint32_t nib = (w >> ((x & 7) * 4)) & 0xFu;
        uint32_t dir;
        if (state == 0) dir = (nib & 1u) ? 3u : 2u;
        else if (state == 1) dir = (nib & 2u) ? 5u : 4u;
        else {
            const uint32_t hs = nib >> 2;
            dir = hs == 0 ? 1u : (hs == 1 ? ((nib & 1u) ? 3u : 2u) : ((nib & 2u) ? 5u : 4u));
        }
        char op;
        switch (dir) {
            case 1: ident += (q[i] == t[j]) ? 1u : 0u; --i; --j; state = 2; op = 'M'; break;
            case 2: --i; state = 0; op = 'I'; break;
            case 3: --i; state = 2; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            default: --j; state = 2; op = 'D'; break;
        }
        out[n++] = op;
    }
    if (!ok || i != 0 || j != 0) {
        info.status = MMGPU_BT_FAILED;
        L.info[ji] = info;
        return;
    }
    // the reference closes the CIGAR with the cell (0,0) as one more 'M' (:1652-1669)
    ident += (q[0] == t[0]) ? 1u : 0u;
    out[n++] = 'M';
    for (uint32_t a = 0, b = n - 1; a < b; a++, b--) {   // reverse in place
        const char c = out[a];
        out[a] = out[b];
        out[b] = c;
    }
    info.bt_len = n;
    info.ident = ident;
    L.info[ji] = info;
#undef AT
#undef DIRW
}

}  // namespace

hipError_t launch_sw_traceback(const BtLaunch &L, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    if (L.band_cap == 0) hipLaunchKernelGGL(sw_traceback_kernel<true>, dim3((L.n_jobs + 63) / 64), dim3(64), 0, stream, L);
    else hipLaunchKernelGGL(sw_traceback_kernel<false>, dim3((L.n_jobs + 63) / 64), dim3(64), 0, stream, L);
    return hipGetLastError();
}

}  // namespace mmgpu
