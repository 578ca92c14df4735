// What sits at an index of a buffer after a sequence of overwrites (integration/MMGpuNuclAlignRun.cpp: the reference's per-thread
// Sequence::numSequence / queryRevCompSeq buffers, of which it reads one residue past the end).  No dependencies: also compiled
// by tests/test_buffer_history.py against a real buffer.
#ifndef MMGPU_BUFFER_HISTORY_H
#define MMGPU_BUFFER_HISTORY_H

#include <cstddef>
#include <vector>

// One buffer of the reference (Sequence::numSequence, BandedNucleotideAligner::queryRevCompSeq) as the list of sequences that
// still show through: chronological, lengths strictly decreasing (a later sequence hides every earlier one that is not longer).
class BufferHistory {
public:
    void map(const unsigned char *seq, size_t len) {
        while (!items.empty() && items.back().len <= len) items.pop_back();
        Item it;
        it.seq = seq;
        it.len = len;
        items.push_back(it);
    }
    // the sequence that owns buffer index `idx` right now, or false: nothing written there yet
    bool owner(size_t idx, const unsigned char **seq, size_t *len) const {
        for (size_t k = items.size(); k-- > 0;)
            if (items[k].len > idx) {
                *seq = items[k].seq;
                *len = items[k].len;
                return true;
            }
        return false;
    }

private:
    struct Item {
        const unsigned char *seq;
        size_t len;
    };
    std::vector<Item> items;
};

#endif
