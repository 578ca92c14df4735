// Resident server mode (the reference's counterpart is src/util/gpuserver.cpp + GPUSharedMemory,
// src/commons/GpuUtil.h:8-52): one process owns the device context with the target database and the k-mer index resident
// in HBM; short-lived `mmseqs prefilter` / `mmseqs align` processes attach to it instead of creating a HIP context and
// uploading the database every time.  The reference exchanges ONE query per hand-shake through a POSIX shared-memory
// segment with an IDLE -> RESERVED -> READY -> DONE state word; here a client sends whole BLOCKS of queries (the unit
// both seams work in) over a unix-domain stream socket, one client at a time (the accept loop is the
// RESERVED state), and a database is identified by a fingerprint so that a client whose database is already resident
// skips the upload.
//
// Wire format: little-endian, every message = WireHdr + payload.  Requests carry the op, replies the status of the
// library call (MMGPU_OK or the C-ABI error code, then the payload is the mmgpu_last_error() text).
#ifndef MMGPU_WIRE_H
#define MMGPU_WIRE_H

#include <errno.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace mmgpu_wire {

constexpr uint32_t MAGIC = 0x4D4D4750u;   // "MMGP"
constexpr uint32_t VERSION = 1;

enum Op : uint32_t {
    OP_HELLO = 1,
    OP_HAS_TARGETS, OP_LOAD_TARGETS, OP_HAS_INDEX, OP_LOAD_INDEX,
    OP_PF_PREPARE, OP_PF_RUN, OP_PF_FETCH, OP_PF_FREE,
    OP_SW_PREPARE, OP_SW_RUN, OP_SW_FETCH, OP_SW_TRACEBACK, OP_SW_FREE,
    OP_BUILD_INDEX, OP_STATS, OP_SHUTDOWN,
    OP_SW_BLOCK_BACKTRACE,     // appended: the ops above keep their numbers
    OP_MASK_TARGETS,           // round 4: mmgpu_pf_mask_targets (tantan on the device)
    OP_SW_REVERSE_PAIRS        // round 6: mmgpu_sw_reverse_pairs (start positions of the pairs the block aligner declined)
};

struct WireHdr {
    uint32_t magic;
    uint32_t op;        // request: Op; reply: echo
    int32_t status;     // reply: return code of the library call
    uint32_t reserved;
    uint64_t len;       // payload bytes
};

struct ServerStats {      // OP_STATS reply
    uint64_t requests, clients, target_uploads, index_uploads, pf_batches, sw_batches;
};

inline bool write_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t w = ::write(fd, c, n > (1u << 30) ? (1u << 30) : n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w;
        n -= (size_t)w;
    }
    return true;
}

inline bool read_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t r = ::read(fd, c, n > (1u << 30) ? (1u << 30) : n);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (r == 0) return false;
        c += r;
        n -= (size_t)r;
    }
    return true;
}

// growing byte buffer with typed append / bounded typed read
struct Buf {
    std::vector<uint8_t> d;
    size_t rd = 0;
    bool bad = false;

    template <typename T> void put(const T &v) {
        const size_t o = d.size();
        d.resize(o + sizeof(T));
        memcpy(d.data() + o, &v, sizeof(T));
    }
    void put_bytes(const void *p, size_t n) {
        put<uint64_t>((uint64_t)n);
        const size_t o = d.size();
        d.resize(o + n);
        if (n) memcpy(d.data() + o, p, n);
    }
    template <typename T> T get() {
        T v;
        memset(&v, 0, sizeof(T));
        if (bad || rd + sizeof(T) > d.size()) {
            bad = true;
            return v;
        }
        memcpy(&v, d.data() + rd, sizeof(T));
        rd += sizeof(T);
        return v;
    }
    // pointer into the buffer (valid while the buffer lives) and the byte count
    const uint8_t *get_bytes(size_t *n) {
        const uint64_t len = get<uint64_t>();
        if (bad || rd + len > d.size()) {
            bad = true;
            *n = 0;
            return nullptr;
        }
        const uint8_t *p = d.data() + rd;
        rd += (size_t)len;
        *n = (size_t)len;
        return len ? p : nullptr;
    }
};

inline bool send_msg(int fd, uint32_t op, int32_t status, const void *payload, size_t len) {
    WireHdr h;
    h.magic = MAGIC;
    h.op = op;
    h.status = status;
    h.reserved = VERSION;
    h.len = len;
    return write_all(fd, &h, sizeof(h)) && (len == 0 || write_all(fd, payload, len));
}

// largest message accepted (targets + index of a large database are a few GB)
static const uint64_t MMGPU_WIRE_MAX_MSG = 1ull << 38;
inline bool recv_msg(int fd, WireHdr *h, Buf *b) {
    if (!read_all(fd, h, sizeof(*h)) || h->magic != MAGIC) return false;
    if (h->len > MMGPU_WIRE_MAX_MSG) return false;      // a length no honest client sends: drop the connection, do not allocate
    b->d.resize((size_t)h->len);
    b->rd = 0;
    b->bad = false;
    return h->len == 0 || read_all(fd, b->d.data(), (size_t)h->len);
}

// 64-bit fingerprint of a byte range (8 bytes per step; not cryptographic - it names a database the way the reference
// names its segment by a hash of the database path, GpuUtil.cpp)
inline uint64_t fingerprint(const void *p, size_t n, uint64_t seed) {
    const uint8_t *c = static_cast<const uint8_t *>(p);
    uint64_t h = seed ^ (0x9E3779B97F4A7C15ull * (n + 1));
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, c + i, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 29;
    }
    uint64_t tail = 0;
    if (i < n) memcpy(&tail, c + i, n - i);
    h = (h ^ tail) * 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 32;
    return h;
}

}  // namespace mmgpu_wire

#endif
