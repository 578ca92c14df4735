// MMGpuBlockBacktracer over the host's own SmithWaterman objects - used by the hook behind Alignment::run (MMGpuAlignRun.cpp).
#ifndef MMGPU_HOST_BLOCK_H
#define MMGPU_HOST_BLOCK_H

#include <string>
#include <vector>

#include "Debug.h"
#include "Sequence.h"
#include "StripedSmithWaterman.h"
#include "SubstitutionMatrix.h"

#include "MMGpuMatcher.h"

// MMGpuBlockBacktracer over the host's own SmithWaterman objects (MMGPU_BLOCK_ALIGNER=host, and the pairs the device's block
// aligner declines as too large): ssw_init once per (thread, query), then the reference's alignStartPosBacktraceBlock through
// the public wrapper the patch adds to SmithWaterman.
class HostBlockBacktracer : public MMGpuBlockBacktracer {
public:
    HostBlockBacktracer(unsigned int threads, size_t maxSeqLen, BaseMatrix *m, bool compBias, float compBiasScale, int gapOpen,
                        int gapExtend, int seqType)
        : m(m), gapOpen(gapOpen), gapExtend(gapExtend), sw(threads, NULL), seq(threads, NULL), lastQuery(threads, (size_t)-1),
          maxSeqLen(maxSeqLen), compBias(compBias), compBiasScale(compBiasScale), seqType(seqType) {
        const int a = m->alphabetSize;
        tiny.resize(a * a);
        for (int i = 0; i < a; i++)
            for (int j = 0; j < a; j++) tiny[i * a + j] = (int8_t)m->subMatrix[i][j];
    }
    ~HostBlockBacktracer() {
        for (size_t i = 0; i < sw.size(); i++) {
            delete sw[i];
            delete seq[i];
        }
    }
    void newBlock() { std::fill(lastQuery.begin(), lastQuery.end(), (size_t)-1); }
    bool run(unsigned int thread, size_t queryIndex, const unsigned char *query, int queryLength, const unsigned char *target,
             int targetLength, s_align &a, std::string &backtrace) {
        if (thread >= sw.size()) {
            Debug(Debug::ERROR) << "MMGPU: OpenMP thread " << thread << " outside the " << sw.size() << " threads of this run\n";
            EXIT(EXIT_FAILURE);
        }
        if (sw[thread] == NULL) {
            sw[thread] = new SmithWaterman(maxSeqLen, m->alphabetSize, compBias, compBiasScale, (SubstitutionMatrix *)m);
            seq[thread] = new Sequence(maxSeqLen, seqType, m, 0, false, compBias);
        }
        if (lastQuery[thread] != queryIndex) {
            seq[thread]->mapSequence(0, 0, std::make_pair(query, (const unsigned int)queryLength));
            sw[thread]->ssw_init(seq[thread], tiny.data(), m);
            lastQuery[thread] = queryIndex;
        }
        s_align r = sw[thread]->mmgpuBlockBacktrace(target, targetLength, (uint8_t)gapOpen, (uint8_t)gapExtend, backtrace, a);
        if (r.score1 == UINT32_MAX) {
            backtrace.clear();
            return false;
        }
        a = r;
        return true;
    }

private:
    BaseMatrix *m;
    int gapOpen, gapExtend;
    std::vector<SmithWaterman *> sw;
    std::vector<Sequence *> seq;
    std::vector<size_t> lastQuery;
    std::vector<int8_t> tiny;
    size_t maxSeqLen;
    bool compBias;
    float compBiasScale;
    int seqType;
};

#endif
