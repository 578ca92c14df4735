// Host check of integration/MMGpuBufferHistory.h (tests/test_buffer_history.py): random sequences are "mapped" into a real buffer
// (memcpy, as Sequence::mapSequence does; a second buffer holds the reverse complement the way BandedNucleotideAligner::initQuery
// writes it) and into the history; after every mapping the letter one past the end, and letters at random indices, must agree.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "MMGpuBufferHistory.h"

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    srand(seed);
    const size_t CAP = 4096;
    std::vector<unsigned char> buffer(CAP + 1, 0), rcBuffer(CAP + 1, 0);      // nothing written yet = 0
    static const unsigned char rev[5] = {2, 3, 0, 1, 4};                       // A C T G X -> T G A C X
    BufferHistory h, hrc;
    std::vector<std::vector<unsigned char> *> keep;
    long checked = 0;
    for (int step = 0; step < 20000; step++) {
        const size_t len = 1 + (size_t)(rand() % (step % 7 == 0 ? CAP : 200));
        std::vector<unsigned char> *s = new std::vector<unsigned char>(len);
        keep.push_back(s);
        for (size_t i = 0; i < len; i++) (*s)[i] = (unsigned char)(rand() % 5);
        memcpy(buffer.data(), s->data(), len);
        for (size_t pos = 0; pos < len; pos++) rcBuffer[(len - 1) - pos] = rev[(*s)[pos]];
        h.map(s->data(), len);
        hrc.map(s->data(), len);
        for (int probe = 0; probe < 4; probe++) {
            const size_t idx = probe == 0 ? len : (size_t)(rand() % (CAP + 1));
            const unsigned char *os;
            size_t ol;
            unsigned got = 0, gotRc = 0;
            if (h.owner(idx, &os, &ol)) got = os[idx];
            if (hrc.owner(idx, &os, &ol)) gotRc = rev[os[ol - 1 - idx]];
            if (got != buffer[idx] || gotRc != rcBuffer[idx]) {
                printf("MISMATCH step %d idx %zu: history %u / %u, buffer %u / %u\n", step, idx, got, gotRc, buffer[idx], rcBuffer[idx]);
                return 1;
            }
            checked++;
        }
    }
    for (size_t i = 0; i < keep.size(); i++) delete keep[i];
    printf("OK %ld letters checked\n", checked);
    return 0;
}
