// Micro-probe: rate at which a wavefront gathers short index lists (8-byte entries, geometric lengths, mean ~3.6 - the lists the
// prefilter's split kernel gathers at 10 000 queries x 1 M targets) out of a 2 GB entry array, in two layouts:
//   packed  the lists back to back (the index as it is: a list of n entries at a random entry offset touches 1 + (n - 1) / 8
//           64-byte lines)
//   padded  a list of up to eight entries that would cross a 64-byte line starts at the next line instead
// Records (start, length) arrive as a sequential stream in random list order, as the similar-k-mer stage leaves them; a wavefront
// takes 64 records, prefix-sums their lengths and reads the entries with lane = entry (adjacent lanes read adjacent entries of a
// list, so the memory side sees one request per line a list touches) - the access pattern of pf_split_kernel's gather without
// its bin arithmetic.  What it answers: is the gather bound by requests (then `padded` is faster by about the ratio of lines
// touched) or by bytes (then it is slower: more bytes for the same entries).
//   hipcc --offload-arch=gfx950 -O3 -o list_gather_rate scripts/probes/list_gather_rate.hip && ./list_gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct Rec { uint32_t start, len; };

// one workgroup = 4 wavefronts, each with its own 64 records per round
__global__ __launch_bounds__(256) void gather(const Rec *recs, uint64_t n_recs, const uint64_t *entries, unsigned long long *sink) {
    __shared__ uint32_t s_pre[4][65], s_start[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    unsigned long long acc = 0;
    for (uint64_t r0 = ((uint64_t)blockIdx.x * 4 + wave) * 64; r0 < n_recs; r0 += waves * 64) {
        const Rec rc = r0 + lane < n_recs ? recs[r0 + lane] : Rec{0, 0};
        // inclusive prefix sum of the lengths over the wavefront
        uint32_t p = rc.len;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(p, d, 64);
            if (lane >= d) p += o;
        }
        s_pre[wave][lane + 1] = p;
        if (lane == 0) s_pre[wave][0] = 0;
        s_start[wave][lane] = rc.start;
        __builtin_amdgcn_wave_barrier();
        const uint32_t total = s_pre[wave][64];
        for (uint32_t t = lane; t < total; t += 64) {
            int lo = 0, hi = 63;      // the list that holds entry t: largest l with pre[l] <= t
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_pre[wave][mid] <= t) lo = mid; else hi = mid - 1;
            }
            acc += entries[(uint64_t)s_start[wave][lo] + (t - s_pre[wave][lo])];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
    const uint64_t n_lists = 60000000;      // ~214 M entries packed = 1.7 GB
    const uint64_t n_recs = 96000000;       // records gathered per launch (lists are drawn with repetition, like k-mers that recur)
    std::vector<uint8_t> len(n_lists);
    std::vector<uint32_t> start_packed(n_lists), start_padded(n_lists);
    uint64_t pos_a = 0, pos_b = 0, lines_a = 0, lines_b = 0, ent = 0;
    for (uint64_t i = 0; i < n_lists; i++) {
        // geometric, P(n) = 0.28 * 0.72^(n - 1): mean 3.57 entries, as ent / sim of the headline workload
        uint32_t n = 1, x = mix((uint32_t)i * 2654435761u + 12345u);
        while (n < 200 && (x & 0xFFFF) >= (uint32_t)(0.28 * 65536)) { n++; x = mix(x + 0x9E3779B9u); }
        len[i] = (uint8_t)n;
        start_packed[i] = (uint32_t)pos_a;
        lines_a += (pos_a + n - 1) / 8 - pos_a / 8 + 1;
        pos_a += n;
        if (n <= 8 && (pos_b & 7) + n > 8) pos_b = (pos_b + 7) & ~7ull;
        start_padded[i] = (uint32_t)pos_b;
        lines_b += (pos_b + n - 1) / 8 - pos_b / 8 + 1;
        pos_b += n;
        ent += n;
    }
    printf("# lists %.3e entries %.3e (mean %.2f); packed %.2f GB, %.3f lines per list; padded %.2f GB (+%.1f %%), %.3f lines per list\n", (double)n_lists,
           (double)ent, (double)ent / n_lists, pos_a * 8 / 1e9, (double)lines_a / n_lists, pos_b * 8 / 1e9, 100.0 * (pos_b - pos_a) / pos_a,
           (double)lines_b / n_lists);
    std::vector<Rec> ra(n_recs), rb(n_recs);
    uint64_t rec_entries = 0;
    for (uint64_t r = 0; r < n_recs; r++) {
        const uint64_t i = ((uint64_t)mix((uint32_t)r * 40503u + 977u) * n_lists) >> 32;
        ra[r] = Rec{start_packed[i], len[i]};
        rb[r] = Rec{start_padded[i], len[i]};
        rec_entries += len[i];
    }
    uint64_t *entries = nullptr;
    unsigned long long *sink = nullptr;
    Rec *d_ra = nullptr, *d_rb = nullptr;
    CHECK(hipMalloc(&entries, (pos_b + 64) * 8));
    CHECK(hipMemset(entries, 1, (pos_b + 64) * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&d_ra, n_recs * sizeof(Rec)));
    CHECK(hipMalloc(&d_rb, n_recs * sizeof(Rec)));
    CHECK(hipMemcpy(d_ra, ra.data(), n_recs * sizeof(Rec), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rb, rb.data(), n_recs * sizeof(Rec), hipMemcpyHostToDevice));
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++)
        for (int layout = 0; layout < 2; layout++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(gather, dim3(cus * 8), dim3(256), 0, 0, layout ? d_rb : d_ra, n_recs, entries, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0)
                printf("%s  records %.3e entries %.3e  ms %8.3f  G lists/s %6.2f  G entries/s %6.2f  G lines/s %6.2f\n", layout ? "padded" : "packed",
                       (double)n_recs, (double)rec_entries, ms, n_recs / ms * 1e-6, rec_entries / ms * 1e-6,
                       (double)(layout ? lines_b : lines_a) / n_lists * n_recs / ms * 1e-6);
        }
    return 0;
}
