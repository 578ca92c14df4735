"""The replay kernel's formulations of CacheFriendlyOperations::findDuplicates' per-target state machine (DESIGN.md section
4.2), as plain Python models on random entry streams - an argument check for the algorithm, not a test of the HIP code
(tests/test_prefilter_gpu.py and scripts/fuzz_prefilter_gpu.py compare the kernels with the oracle on the device):

  sequential   per target: prev = low byte of the previous entry's diagonal (0 at the start); an entry whose byte equals prev
               is flagged; flagged entries are run-length de-duplicated per target on that byte (an entry is emitted unless the
               target's last emitted byte is the same)
  rounds       the kernel's form: 64 entries at a time; inside a round the entries of one target are resolved against each
               other (predecessor's byte; runs of flagged entries), the round's last entry of a target writes the state
  compact      state = prev byte + "has emitted" bit per target, the last emitted byte only for targets that have emitted, in
               a table of at most `cap` entries; a bucket that needs more is redone with the full state
  shortcut     a round in which no target occurs twice skips the matching
"""
import numpy as np
import pytest


def sequential(keys, d8):
    prev, last, emitted = {}, {}, set()
    out = []
    for i, (k, d) in enumerate(zip(keys, d8)):
        flag = d == prev.get(k, 0)
        prev[k] = d
        if flag and (k not in emitted or last[k] != d):
            out.append(i)
        if flag:
            emitted.add(k)
            last[k] = d
    return out


def rounds(keys, d8, cap=None, shortcut=True):
    """returns (emitted entry indices, None) or (None, 'redo') when the compact table overflows"""
    prev, last_full, emitted = {}, {}, set()
    table = {}          # compact: key -> last emitted byte, at most cap entries
    out = []
    for r0 in range(0, len(keys), 64):
        ks, ds = keys[r0:r0 + 64], d8[r0:r0 + 64]
        n = len(ks)
        uniq = shortcut and len(set(ks)) == n
        keep = [False] * n
        flag = [False] * n
        for l in range(n):
            k = ks[l]
            if uniq:
                pl = None
            else:
                pl = max((m for m in range(l) if ks[m] == k), default=None)          # predecessor inside the round
            prevd = ds[pl] if pl is not None else prev.get(k, 0)
            flag[l] = ds[l] == prevd
        for l in range(n):
            if not flag[l]:
                continue
            k = ks[l]
            fbelow = None if uniq else max((m for m in range(l) if ks[m] == k and flag[m]), default=None)
            if fbelow is not None:
                keep[l] = ds[fbelow] != ds[l]
            else:
                em = k in emitted
                le = (table[k] if cap is not None else last_full[k]) if em else None
                keep[l] = (not em) or le != ds[l]
        # state update by the last entry of every target in the round
        fresh = []
        for l in range(n):
            k = ks[l]
            if not uniq and any(ks[m] == k for m in range(l + 1, n)):
                continue
            prev[k] = ds[l]
            fm = [m for m in range(n) if ks[m] == k and flag[m]]
            if fm:
                if k not in emitted:
                    fresh.append(k)
                (table if cap is not None else last_full)[k] = ds[fm[-1]]
        if cap is not None and len(table) > cap:
            return None, "redo"
        emitted.update(fresh)
        out.extend(r0 + l for l in range(n) if keep[l])
    return out, None


@pytest.mark.parametrize("seed", range(6))
def test_round_form_and_compact_state_equal_the_sequential_state_machine(seed):
    rng = np.random.default_rng(seed)
    for trial in range(40):
        n = int(rng.integers(1, 700))
        n_keys = int(rng.choice([3, 40, 500, 4096]))
        n_diag = int(rng.choice([2, 5, 256]))
        keys = rng.integers(0, n_keys, n).tolist()
        d8 = rng.integers(0, n_diag, n).tolist()
        if trial % 3 == 0:          # homolog-like: runs of one target on one diagonal, interleaved with noise
            for _ in range(int(rng.integers(1, 6))):
                k, d = int(rng.integers(0, n_keys)), int(rng.integers(0, n_diag))
                for p in rng.choice(n, min(n, int(rng.integers(2, 30))), replace=False):
                    keys[p], d8[p] = k, d
        want = sequential(keys, d8)
        for shortcut in (False, True):
            got, _ = rounds(keys, d8, cap=None, shortcut=shortcut)
            assert got == want
            for cap in (64, 2, 0):
                got, redo = rounds(keys, d8, cap=cap, shortcut=shortcut)
                if redo:      # more emitting targets than the table holds: the full form takes over (pf_replay_redo_kernel)
                    got, _ = rounds(keys, d8, cap=None, shortcut=shortcut)
                assert got == want
