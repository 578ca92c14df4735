// What sits at an index of a buffer after a sequence of overwrites (integration/MMGpuNuclAlignRun.cpp: the reference's per-thread
// Sequence::numSequence / queryRevCompSeq buffers, of which it reads one residue past the end).  No dependencies: also compiled
// by tests/test_buffer_history.py against a real buffer.
#ifndef MMGPU_BUFFER_HISTORY_H
#define MMGPU_BUFFER_HISTORY_H

#include <cstddef>
#include <vector>

// One buffer of the reference (Sequence::numSequence, BandedNucleotideAligner::queryRevCompSeq) as the list of sequences that
// still show through: chronological, lengths strictly decreasing (a later sequence hides every earlier one that is not longer).
// keepCopies: the history owns a copy of every sequence it still lists (a handful: the lengths decrease), so the caller may free
// its sequences - the hook's query histories, whose sequences come bucket by bucket; otherwise it keeps the caller's pointers
// (the target history: the targets stay resident for the whole run).
class BufferHistory {
public:
    explicit BufferHistory(bool keepCopies = false) : keepCopies(keepCopies) {}
    void map(const unsigned char *seq, size_t len) {
        while (!items.empty() && items.back().len <= len) items.pop_back();
        items.push_back(Item());
        Item &it = items.back();
        if (keepCopies) {
            it.copy.assign(seq, seq + len);
            it.seq = it.copy.data();
        } else {
            it.seq = seq;
        }
        it.len = len;
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
        std::vector<unsigned char> copy;
        Item() : seq(NULL), len(0) {}
        Item(const Item &o) : seq(o.seq), len(o.len), copy(o.copy) { if (!copy.empty()) seq = copy.data(); }
        Item &operator=(const Item &o) {
            copy = o.copy;
            len = o.len;
            seq = copy.empty() ? o.seq : copy.data();
            return *this;
        }
    };
    bool keepCopies;
    std::vector<Item> items;
};

#endif
