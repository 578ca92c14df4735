// Banded traceback (CIGAR), one WAVEFRONT per alignment: SmithWaterman::banded_sw + its traceback
// (src/alignment/StripedSmithWaterman.cpp:1478-1693, SEQ_SEQ branch) and the identity count of computerBacktrace
// (:1280-1308), bit for bit - the default kernel behind mmgpu_sw_traceback; bt_kernel.hip (one lane per alignment, the
// reference's arrays kept literally) serves the bands this one declines.
//
// The reference walks a row of the band cell by cell because F chains along the row: f[j] = max(H[j-1] - go, f[j-1] - ge).
// Two restatements make a row data-parallel (both checked against the oracle's literal restatement over 800 random
// alignments incl. wide and clipped bands, scripts/bt_formulation_check.py, and by the GPU parity tests):
//   * F from "H without its F term": with Hnf = max(max(E,0), diagonal), f[j] = max(Hnf[j-1] - go, f[j-1] - ge) has the
//     same value AND the same tie flag (temp1 > temp2, :1551-1557) as the reference's expression on the full H, because
//     gap_open >= gap_extend and Hnf >= 0.  f is then a max-plus prefix scan: f[l] = max(P[l] - (l-1) ge, f_in - l ge) with
//     P = exclusive prefix maximum of Hnf[m] - go + m ge over the lanes of a chunk - one DPP scan instead of a serial chain;
//   * previous-row values BY COLUMN instead of the reference's band-frame arrays: H / E of row i - 1 live in a ring in LDS
//     indexed by the target column; what the frame arrays' zeroed slots (h_b[0], h_b[edge], :1528) amount to is spelled out:
//     columns outside the previous row's band read 0, and the previous-row H / E seen at the LAST column of row i are 0
//     when i <= band + 1 or the band is not clipped by the target end (`zero_last`; in the clipped rows with i <= band + 1
//     this destroys a valid cell - the reference's quirk, kept).
// Lanes = 64 consecutive columns of the row (wider bands: chunks with carries), rows in sequence; per row-chunk: three LDS
// reads, a 6-step DPP scan, two wave shifts, the direction byte.  The band is doubled until the banded maximum reaches the
// Smith-Waterman score like the reference does; the passes that search for the band run without direction storage, the
// final one is repeated with it (the size of the direction matrix is then known: qlen x (2 band + 1) bytes from a bump
// allocator over one scratch pool).  The walk back is serial (lane 0).
// Bound: issue / LDS latency of ~100 instructions per row-chunk; 34 000 alignments of a 1000-query hit-list batch take a
// few milliseconds where the lane-per-alignment kernel took 200.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr int BTW_RING = 1024;           // columns per wave in LDS: 2 * band + 2 <= BTW_RING
constexpr int BTW_NEG = -(1 << 29);

#define BTW_SCAN_STEP(v, ctrl, rmask)                                                          \
    do {                                                                                       \
        const int o__ = __builtin_amdgcn_update_dpp(BTW_NEG, (v), (ctrl), (rmask), 0xF, false); \
        (v) = (v) > o__ ? (v) : o__;                                                           \
    } while (0)

// inclusive prefix maximum over the 64 lanes (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31)
__device__ __forceinline__ int btw_scan_max(int v) {
    BTW_SCAN_STEP(v, 0x111, 0xF);
    BTW_SCAN_STEP(v, 0x112, 0xF);
    BTW_SCAN_STEP(v, 0x114, 0xF);
    BTW_SCAN_STEP(v, 0x118, 0xF);
    BTW_SCAN_STEP(v, 0x142, 0xA);
    BTW_SCAN_STEP(v, 0x143, 0xC);
    return v;
}
// value of lane - 1; lane 0 receives `first`
__device__ __forceinline__ int btw_shift_up(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// One pass of the banded DP at half-width bw.  WRITE: direction bytes to dir[i * width_d + (j - shift_i)].
template <bool WRITE>
__device__ __forceinline__ int btw_pass(const uint8_t *q, const int8_t *cb, const uint8_t *t, int ql, int tl, int bw, int go, int ge,
                                        int alph, const int8_t *smat, int *Hring, int *Ering, uint8_t *dir,
                                        const int8_t *prof, int qfull) {
    const int lane = (int)(threadIdx.x & 63u);
    const int width_d = 2 * bw + 1;
    int lane_max = 0;
    for (int i = 0; i < ql; i++) {
        const int beg = max(0, i - bw), end = min(tl - 1, i + bw);
        const int pbeg = max(0, i - 1 - bw), pend = min(tl - 1, i - 1 + bw);
        const bool zero_last = i <= bw + 1 || i + bw <= tl - 1;
        const int qi = (int)q[i] * alph;
        const int cbi = (int)cb[i];
        const int sh_i = max(0, i - bw);
        int f_carry = 0, hnf_carry = 0;       // virtual predecessor of the row's first column: H = 0, f = 0
        int hd_carry = (i > 0 && beg - 1 >= pbeg && beg - 1 <= pend) ? Hring[(beg - 1) & (BTW_RING - 1)] : 0;
        for (int j0 = beg; j0 <= end; j0 += 64) {
            const int j = j0 + lane;
            const bool act = j <= end;
            const bool in_prev = act && i > 0 && j >= pbeg && j <= pend;
            const int hold = in_prev ? Hring[j & (BTW_RING - 1)] : 0;
            const int eold = in_prev ? Ering[j & (BTW_RING - 1)] : 0;
            const bool zl = zero_last && j == end;
            const int hp = zl ? 0 : hold, ep = zl ? 0 : eold;
            const int hd = btw_shift_up(hold, hd_carry);
            hd_carry = __builtin_amdgcn_readlane(hold, 63);
            const int te1 = i == 0 ? -go : hp - go, te2 = i == 0 ? -ge : ep - ge;
            const int ev = te1 > te2 ? te1 : te2;
            const unsigned de = te1 > te2 ? 1u : 0u;
            const int e1 = ev > 0 ? ev : 0;
            const int tj = act ? (int)t[j] : 0;
            // profile query: prof points at row q_start of the letter-major profile, qfull = its row length (:1565-1567)
            const int diag = hd + (prof ? (int)prof[tj * qfull + i] : (int)smat[qi + tj] + cbi);
            const int hnf = e1 > diag ? e1 : diag;
            // F: max-plus prefix scan over the chunk
            const int f_in = max(hnf_carry - go, f_carry - ge);
            const int x = act ? hnf - go + lane * ge : BTW_NEG;
            const int incl = btw_scan_max(x);
            const int excl = btw_shift_up(incl, BTW_NEG);
            int f = max(excl - (lane - 1) * ge, f_in - lane * ge);
            if (lane == 0) f = f_in;
            const int hnf_prev = btw_shift_up(hnf, hnf_carry), f_prev = btw_shift_up(f, f_carry);
            const unsigned df = (hnf_prev - go > f_prev - ge) ? 1u : 0u;
            const int f1 = f > 0 ? f : 0;
            const int a = e1 > f1 ? e1 : f1;
            const int h = a > diag ? a : diag;
            if (act) {
                Hring[j & (BTW_RING - 1)] = h;
                Ering[j & (BTW_RING - 1)] = ev;
                lane_max = h > lane_max ? h : lane_max;
                if (WRITE) {
                    const unsigned hsel = a <= diag ? 0u : (e1 > f1 ? 1u : 2u);
                    dir[(size_t)i * (size_t)width_d + (size_t)(j - sh_i)] = (uint8_t)(de | (df << 1) | (hsel << 2));
                }
            }
            // carries into the next chunk: the last lane's values (a chunk that is not the row's last is full)
            f_carry = __builtin_amdgcn_readlane(f, 63);
            hnf_carry = __builtin_amdgcn_readlane(hnf, 63);
        }
    }
    // maximum over the lanes
    int v = lane_max;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void sw_traceback_wave_kernel(BtLaunch L) {
    __shared__ int8_t smat[32 * 32];
    __shared__ int s_h[4][BTW_RING], s_e[4][BTW_RING];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256) smat[k] = k < L.alphabet * L.alphabet ? L.mat[k] : (int8_t)0;
    __syncthreads();
    const uint32_t jidx = blockIdx.x * 4u + (uint32_t)wave;
    if (jidx >= L.n_jobs) return;
    const BtJob J = L.jobs[jidx];
    int *Hring = s_h[wave], *Ering = s_e[wave];
    const int ql = J.q_end - J.q_start + 1, tl = J.t_end - J.t_start + 1;
    const uint8_t *q = L.q_res + L.q_off[J.query] + J.q_start;
    const int8_t *cb = L.q_cb + L.q_off[J.query] + J.q_start;
    const uint8_t *t = L.t_res + (size_t)L.t_off4[J.target] * 4 + J.t_start;
    const int go = L.gap_open, ge = L.gap_extend, alph = L.alphabet;
    const int8_t *prof = nullptr;
    int qfull = 0;
    if (L.q_prof_off && L.q_prof_off[J.query] != 0xFFFFFFFFu) {
        qfull = (int)(L.q_off[J.query + 1] - L.q_off[J.query]);
        prof = L.q_prof + L.q_prof_off[J.query] + J.q_start;
    }
    mmgpu_sw_bt info;
    info.bt_off = J.bt_off;
    info.bt_len = 0;
    info.ident = 0;
    info.status = MMGPU_BT_OK;
    info.reserved = 0;

    // ---- the band: |tlen - qlen| + 1, doubled until the banded maximum reaches the score (:1500-1588)
    int bw = (tl > ql ? tl - ql : ql - tl) + 1;
    bool fail = false;
    for (;;) {
        if (2 * bw + 2 > BTW_RING) { fail = true; break; }
        const int mx = btw_pass<false>(q, cb, t, ql, tl, bw, go, ge, alph, smat, Hring, Ering, nullptr, prof, qfull);
        if (mx >= J.score) break;
        bw *= 2;
    }
    unsigned long long off = 0;
    const unsigned long long need = (unsigned long long)ql * (unsigned long long)(2 * bw + 1);
    if (!fail) {
        if (lane == 0) off = atomicAdd(L.dir_cursor, need);
        off = (unsigned long long)__shfl((long long)off, 0, 64);
        if (off + need > L.dir_pool_bytes) fail = true;
    }
    if (fail) {      // band or direction storage beyond this kernel: the lane-per-alignment kernel (or the host) takes the pair
        if (lane == 0) {
            info.status = MMGPU_BT_TOO_LARGE;
            L.info[J.slot] = info;
        }
        return;
    }
    uint8_t *dir = L.dir_pool + off;
    (void)btw_pass<true>(q, cb, t, ql, tl, bw, go, ge, alph, smat, Hring, Ering, dir, prof, qfull);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // the direction bytes were written by all lanes
    __builtin_amdgcn_wave_barrier();

    // ---- traceback from the bottom-right corner in state H (:1590-1651); ops are produced last to first
    char *out = L.bt + J.bt_off;
    const uint32_t cap = (uint32_t)(ql + tl + 1);
    const int width_d = 2 * bw + 1;
    uint32_t n = 0, ident = 0;
    int ok = 1;
    if (lane == 0) {
        int i = ql - 1, j = tl - 1, state = 2;
        while (i > 0 || j > 0) {
            const int sh = max(0, i - bw);
            const int x = j - sh;
            if (i < 0 || j < 0 || x < 0 || j > min(tl - 1, i + bw) || n + 1 >= cap) {
                ok = 0;   // the walk left the band: the reference would read unrelated direction bytes here
                break;
            }
            const uint32_t nib = dir[(size_t)i * (size_t)width_d + (size_t)x];
            uint32_t d;
            if (state == 0) d = (nib & 1u) ? 3u : 2u;
            else if (state == 1) d = (nib & 2u) ? 5u : 4u;
            else {
                const uint32_t hs = nib >> 2;
                d = hs == 0 ? 1u : (hs == 1 ? ((nib & 1u) ? 3u : 2u) : ((nib & 2u) ? 5u : 4u));
            }
            char op;
            switch (d) {
                case 1: ident += (q[i] == t[j]) ? 1u : 0u; --i; --j; state = 2; op = 'M'; break;
                case 2: --i; state = 0; op = 'I'; break;
                case 3: --i; state = 2; op = 'I'; break;
                case 4: --j; state = 1; op = 'D'; break;
                default: --j; state = 2; op = 'D'; break;
            }
            out[n++] = op;
        }
        if (ok && (i != 0 || j != 0)) ok = 0;
        if (ok) {
            // the reference closes the CIGAR with the cell (0,0) as one more 'M' (:1652-1669)
            ident += (q[0] == t[0]) ? 1u : 0u;
            out[n++] = 'M';
        }
    }
    ok = __shfl(ok, 0, 64);
    n = (uint32_t)__shfl((int)n, 0, 64);
    if (!ok) {
        if (lane == 0) {
            info.status = MMGPU_BT_FAILED;
            L.info[J.slot] = info;
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // reverse in place, all lanes
    for (uint32_t a = (uint32_t)lane; a < n / 2; a += 64) {
        const uint32_t b = n - 1 - a;
        const char c = out[a];
        out[a] = out[b];
        out[b] = c;
    }
    if (lane == 0) {
        info.bt_len = n;
        info.ident = ident;
        L.info[J.slot] = info;
    }
}

}  // namespace

hipError_t launch_sw_traceback_wave(const BtLaunch &L, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_traceback_wave_kernel, dim3((L.n_jobs + 3) / 4), dim3(256), 0, stream, L);
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_bt() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_traceback_wave_kernel));
}

}  // namespace mmgpu
