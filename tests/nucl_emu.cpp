// TEST INFRASTRUCTURE - host emulation of the 16 lanes of one alignment group, so that the very source of the GPU
// kernel (mmseqs2_amd/csrc/nucl_core.h) can be run against the reference's vectors on a machine without a GPU.
// 16 cooperative contexts (ucontext) stand in for the lanes; a context switch happens exactly where the kernel marks a
// phase boundary (NUCL_SYNC) or exchanges a value inside the group, which is where lock-step lanes would see each
// other's LDS writes.  Built and loaded by tests/test_nucl_emu.py only; nothing in the product links this.
#include <ucontext.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace emu {
#ifndef EMU_LANES
#define EMU_LANES 16
#endif
constexpr int LANES = EMU_LANES;
static ucontext_t ctx[LANES], main_ctx;
static bool done[LANES];
static int cur = 0, arrived = 0;
static unsigned generation = 0;
static unsigned long long xchg[LANES];

static void switch_to_next() {
    int from = cur;
    for (int k = 1; k <= LANES; k++) {
        const int n = (from + k) % LANES;
        if (!done[n]) {
            if (n == from) return;
            cur = n;
            swapcontext(&ctx[from], &ctx[n]);
            return;
        }
    }
    swapcontext(&ctx[from], &main_ctx);   // every lane has finished
}

static void sync() {
    const unsigned gen = generation;
    if (++arrived == LANES) {
        arrived = 0;
        generation++;
    } else {
        while (generation == gen) switch_to_next();
    }
}

static unsigned long long exchange(unsigned long long v, int src) {
    xchg[cur] = v;
    sync();
    const unsigned long long r = xchg[src & (LANES - 1)];
    sync();
    return r;
}
}  // namespace emu

#define NUCL_NG EMU_LANES
#define NUCL_HD inline
#define NUCL_LANE() (emu::cur)
template <typename T> static inline T emu_shfl(T v, int src) { return (T)emu::exchange((unsigned long long)(long long)v, src); }
#define NUCL_SHFL(v, src) emu_shfl((v), (src))
#define NUCL_SHFL_XOR(v, mask) emu_shfl((v), emu::cur ^ (mask))
#define NUCL_SHFL_U64(v, src) (emu::exchange((v), (src)))
#define NUCL_SYNC() emu::sync()
#define NUCL_SYNC_MEM() emu::sync()
static inline unsigned emu_add32(uint32_t *p, unsigned v) { const unsigned o = *p; *p += v; return o; }
static inline unsigned long long emu_add64(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
#define NUCL_ATOMIC_ADD_U32(p, v) emu_add32((p), (v))
#define NUCL_ATOMIC_ADD_U64(p, v) emu_add64((p), (v))
#include "nucl_core.h"
#ifdef EMU_WAVE   // the one-wavefront-per-alignment kernel (nucl_wave.h): 64 emulated lanes, its two extra primitives
#define NUCL_ROR1_U32(v) ((unsigned)emu::exchange((unsigned long long)(unsigned)(v), (emu::cur + emu::LANES - 1) % emu::LANES))
#define NUCL_READLANE(v, l) ((int)emu::exchange((unsigned long long)(long long)(int)(v), (l)))
static inline int emu_wave_max(int v) {
    for (int d = 1; d < emu::LANES; d <<= 1) { const int o = (int)emu::exchange((unsigned long long)(long long)v, emu::cur ^ d); v = o > v ? o : v; }
    return v;
}
static inline unsigned emu_wave_minu(unsigned v) {
    for (int d = 1; d < emu::LANES; d <<= 1) { const unsigned o = (unsigned)emu::exchange((unsigned long long)v, emu::cur ^ d); v = o < v ? o : v; }
    return v;
}
#define NUCL_WAVE_MAX_I32(v) emu_wave_max(v)
#define NUCL_WAVE_MIN_U32(v) emu_wave_minu(v)
#include "nucl_wave.h"
#endif

namespace {
const mmgpu::NuclLaunch *g_launch;
#ifdef EMU_WAVE
mmgpu::nuclw::WaveLds g_lds;
#else
mmgpu::NUCL_NS::GroupLds g_lds;
#endif
uint8_t *g_p;
char *g_w;

void lane_main() {
#ifdef EMU_WAVE
    mmgpu::nuclw::align_wave(*g_launch, g_lds, g_p, g_w);
#else
    mmgpu::NUCL_NS::align_group(*g_launch, g_lds, g_p, g_w);
#endif
    emu::done[emu::cur] = true;
    emu::switch_to_next();
}
}  // namespace

// Same contract as mmgpu_nucl_align (include/mmgpu.h), with the target database passed explicitly.
extern "C" int nucl_emu_align(const mmgpu_nucl_params *par, const mmgpu_nucl_query *qs, uint32_t nq, const uint8_t *t_res,
                              const uint64_t *t_off, uint32_t n_targets, const mmgpu_nucl_pair *pairs, uint32_t n_pairs,
                              mmgpu_nucl_hit *out, char *bt, uint64_t bt_cap, uint64_t *bt_used) {
    std::vector<uint32_t> qoff(nq + 1, 0);
    for (uint32_t i = 0; i < nq; i++) qoff[i + 1] = qoff[i] + qs[i].qlen;
    std::vector<uint8_t> qres(qoff[nq] + 16);
    for (uint32_t i = 0; i < nq; i++) memcpy(qres.data() + qoff[i], qs[i].q, qs[i].qlen);
    std::vector<uint32_t> off4(n_targets), len(n_targets), order(n_pairs);
    uint64_t cur4 = 0, longest = 0;
    for (uint32_t i = 0; i < n_targets; i++) {
        off4[i] = (uint32_t)cur4;
        len[i] = (uint32_t)(t_off[i + 1] - t_off[i]);
        cur4 += (len[i] + 3) / 4;
    }
    std::vector<uint8_t> tres(cur4 * 4 + 16, 5);
    for (uint32_t i = 0; i < n_targets; i++) memcpy(tres.data() + (size_t)off4[i] * 4, t_res + t_off[i], len[i]);
    for (uint32_t i = 0; i < n_pairs; i++) {
        order[i] = i;
        const uint64_t span = (uint64_t)qs[pairs[i].query].qlen + len[pairs[i].target];
        longest = span > longest ? span : longest;
    }
    std::vector<uint8_t> pbuf((longest * 6 + 2) * 16);
    std::vector<char> wbuf(longest + 32);
    unsigned long long counters[2] = {0, 0};
    mmgpu::NuclLaunch L;
    L.pairs = pairs;
    L.order = order.data();
    L.n_pairs = n_pairs;
    L.q_res = qres.data();
    L.q_off = qoff.data();
    L.t_res = tres.data();
    L.t_off4 = off4.data();
    L.t_len = len.data();
    for (int i = 0; i < 25; i++) L.mat[i] = par->mat[i];
    for (int i = 0; i < 8; i++) L.rev_lookup[i] = i < 5 ? par->reverse[i] : (uint8_t)4;
    L.gapo = par->gap_open;
    L.gape = par->gap_extend;
    L.zdrop = par->zdrop;
    L.wrapped = par->wrapped ? 1 : 0;
    L.past_end_q = par->past_end_query;
    L.past_end_t = par->past_end_target;
    L.pscratch = pbuf.data();
    L.pscratch_stride = pbuf.size();
    L.wscratch = wbuf.data();
    L.wscratch_stride = wbuf.size();
    L.out = out;
    L.bt = bt;
    L.bt_cursor = &counters[0];
    L.bt_cap = bt_cap;
    L.next_pair = reinterpret_cast<uint32_t *>(&counters[1]);
    g_launch = &L;
    g_p = pbuf.data();
    g_w = wbuf.data();
    const size_t stack = 1u << 18;
    std::vector<char> stacks(stack * emu::LANES);
    emu::arrived = 0;
    emu::generation = 0;
    for (int l = 0; l < emu::LANES; l++) {
        emu::done[l] = false;
        getcontext(&emu::ctx[l]);
        emu::ctx[l].uc_stack.ss_sp = stacks.data() + stack * l;
        emu::ctx[l].uc_stack.ss_size = stack;
        emu::ctx[l].uc_link = &emu::main_ctx;
        makecontext(&emu::ctx[l], lane_main, 0);
    }
    emu::cur = 0;
    swapcontext(&emu::main_ctx, &emu::ctx[0]);
    if (bt_used) *bt_used = counters[0];
    return 0;
}
