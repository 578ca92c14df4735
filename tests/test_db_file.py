"""Persisted device layout (include/mmgpu.h mmgpu_db_save / mmgpu_db_probe / mmgpu_db_load): a context
brought back from the file answers prefilter and alignment calls exactly like the context the file was written from, a file of
another database or with other index parameters is refused (the caller builds as ever), and the header can be read without a
device."""
import numpy as np
import pytest

from mmseqs2_amd import capi, workloads as wl
from tests import pf_common as pc


def test_probe_refuses_what_is_not_a_database_file(tmp_path):
    import torch  # noqa: F401  (before libmmgpu.so, as in conftest.gpu: the process must end up with one HIP runtime)
    assert capi.db_probe(tmp_path / "missing.mmgpu") is None
    junk = tmp_path / "junk.mmgpu"
    junk.write_bytes(b"MMGPUDB1" + b"\0" * 100)
    assert capi.db_probe(junk) is None
    # a header of this version whose sections lie inside the file is accepted, the same header with any section (or its size)
    # reaching past the end of the file is not - before a byte of it is sent to the device
    import struct
    fmt = "<8sIIQQIIIIQQIIiiiiQQ7Q6Q"
    n, res_bytes, table, n_entries = 10, 8192, 21 ** 6, 100
    at = [4096, 8192, 12288, 20480, 28672, 28672 + ((4 * (table + 1) + 4095) // 4096) * 4096]
    file_bytes = at[5] + 4096

    def header(**over):
        f = dict(version=2, n=n, res_bytes=res_bytes, table=table, n_entries=n_entries, at=list(at), file_bytes=file_bytes)
        f.update(over)
        return struct.pack(fmt, b"MMGPUDB1", f["version"], struct.calcsize(fmt), 0x11, 0x22, f["n"], 21, 500, 300, 3000, f["res_bytes"], 1, 1,
                           6, 1, 21, 0, f["table"], f["n_entries"], *f["at"], f["file_bytes"], *([0] * 6))

    def probe(h):
        q = tmp_path / "crafted.mmgpu"
        q.write_bytes(h + b"\0" * (file_bytes - len(h)))
        return capi.db_probe(q)
    ok = probe(header())
    assert ok is not None and ok["n_targets"] == n and ok["n_entries"] == n_entries and ok["kmer_size"] == 6
    assert probe(header(version=1)) is None                                         # files of the previous layout: rebuilt, not read
    for k in range(6):
        moved = list(at)
        moved[k] = file_bytes - 8
        assert probe(header(at=moved)) is None, k
    assert probe(header(res_bytes=file_bytes)) is None
    assert probe(header(n_entries=2 ** 61)) is None and probe(header(table=2 ** 62)) is None      # (sizes that would wrap the sums)
    assert probe(header(n=2 ** 31)) is None
    assert probe(header(file_bytes=file_bytes + 1)) is None                         # truncated file


@pytest.mark.gpu
def test_saved_database_answers_like_the_resident_one(gpu, matrices, oracle, tmp_path):
    import mmseqs2_amd
    g = pc.golden()
    km16, um8 = g["vtml80_kmer16"], g["blosum62_ungapped"]
    thr = int(g["kmer_thr"])
    tv = np.load(pc.GOLDEN + "/tantan_vectors.npz")
    s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
    gpu.load_targets(g["tres"], g["toff"], 21)
    n_masked = gpu.pf_mask_targets(tv["vtml80_likelihood_ratios"], float(tv["mask_prob"]), 20)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    qs = pc.golden_queries(g)
    for qd in qs:
        qd["identity_id"] = None
    hits_a, counts_a, status_a, _ = gpu.pf_batch(qs, thr, max_hits=300, ref_bins=2)
    off_a, ids_a, pos_a = gpu.pf_debug_index(6, 21)
    path = tmp_path / "golden.mmgpu"
    gpu.db_save(path, source_fp=0x1234, index_fp=0x77)
    info = capi.db_probe(path, gpu.L)
    assert info is not None and info["n_targets"] == len(g["toff"]) - 1 and info["has_index"] == 1 and info["kmer_size"] == 6
    assert info["has_masked_view"] == (1 if n_masked >= 0 else 0) and info["source_fingerprint"] == 0x1234 and info["index_fingerprint"] == 0x77
    assert info["n_entries"] == len(ids_a) and info["total_residues"] == int(g["toff"][-1])

    other = mmseqs2_amd.MMGpu(0)
    try:
        # another database / other index parameters: refused, the context stays as it was (empty)
        assert other.db_load(path, 0x9999, 0x77, 6, 21, True, s3, i3, um8) is False
        assert other.db_load(path, 0x1234, 0x78, 6, 21, True, s3, i3, um8) is False
        assert other.db_load(tmp_path / "missing.mmgpu", 0x1234, 0x77, 6, 21, True, s3, i3, um8) is False
        # the real thing: no sequence handed over, no masking, no index build
        assert other.db_load(path, 0x1234, 0x77, 6, 21, True, s3, i3, um8) is True
        off_b, ids_b, pos_b = other.pf_debug_index(6, 21)
        assert np.array_equal(off_a, off_b) and np.array_equal(ids_a, ids_b) and np.array_equal(pos_a, pos_b)
        assert np.array_equal(other.pf_debug_masked_targets(g["toff"]), gpu.pf_debug_masked_targets(g["toff"]))
        hits_b, counts_b, status_b, _ = other.pf_batch(qs, thr, max_hits=300, ref_bins=2)
        assert np.array_equal(counts_a, counts_b) and np.array_equal(status_a, status_b)
        for qi in range(len(qs)):
            n = int(counts_a[qi])
            assert np.array_equal(hits_a[qi][:n], hits_b[qi][:n])
        # the alignment kernels read the UNMASKED residues of the file
        mat = matrices["blosum62_sw"]
        q = qs[0]["q"]
        ids = np.arange(40, dtype=np.uint32)
        sw_q = [dict(q=q, comp_bias=None, targets=ids, min_start_score=0)]
        a = gpu.sw_batch(mat, 11, 1, sw_q, mode=1)
        b = other.sw_batch(mat, 11, 1, sw_q, mode=1)
        assert np.array_equal(a, b)
        tl = wl.split(g["tres"], g["toff"])
        for k in (0, 7, 39):
            r = oracle.sw_align(q, None, tl[k], mat, 11, 1, need_start=True)
            assert (int(b[k]["score"]), int(b[k]["q_end"]), int(b[k]["t_end"])) == (r["score"], r["q_end"], r["t_end"])
        # targets only (what an alignment module asks for): no index afterwards
        assert other.db_load(path, 0x1234, 0) is True
        assert np.array_equal(other.sw_batch(mat, 11, 1, sw_q, mode=1), a)
        with pytest.raises(capi.MMGpuError):
            other.pf_batch(qs[:1], thr, max_hits=300, ref_bins=2)
        # damaged files are refused and leave the context as it was: a flipped byte in the offset table, in the residues, a
        # truncated file, a header whose section lies outside the file (the resident targets keep answering)
        import struct
        raw = bytearray(path.read_bytes())
        fields = struct.unpack_from("<8sIIQQIIIIQQIIiiiiQQ7Q6Q", raw, 0)
        at_off4, at_len, at_res, at_masked, at_offsets, at_entries, file_bytes = fields[19:26]
        assert file_bytes == len(raw) and at_off4 < at_len < at_res < at_masked < at_offsets < at_entries
        def damaged(name, change):
            bad = bytearray(raw)
            bad = change(bad) or bad
            q = tmp_path / name
            q.write_bytes(bytes(bad))
            return q
        def flip(at):
            def f(bb):
                bb[at] ^= 0x40
            return f
        cases = {"offsets": flip(at_offsets + 4 * 1000 + 1), "residues": flip(at_res + 777), "entries": flip(at_entries + 8 * 50),
                 "lengths": flip(at_len + 9), "truncated": lambda bb: bb[:at_entries + 100],
                 "section_outside": lambda bb: struct.pack_into("<Q", bb, 8 + 8 + 16 + 16 + 16 + 8 + 16 + 16 + 4 * 8, len(raw) + 4096)}
        for name, change in cases.items():
            q = damaged(name + ".mmgpu", change)
            assert other.db_load(q, 0x1234, 0x77, 6, 21, True, s3, i3, um8) is False, name
            assert np.array_equal(other.sw_batch(mat, 11, 1, sw_q, mode=1), a), name
    finally:
        other.close()
    # restore the module's golden case for the tests that follow
    from tests import pf_gpu_check as chk
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)
