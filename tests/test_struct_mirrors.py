"""The ctypes mirrors of the C structs (mmseqs2_amd/capi.py for include/mmgpu.h, oracle/pyoracle.py for oracle/mm_oracle.h) must have
the C compiler's sizes: a field added on one side only makes the C side read past the Python object - wrong answers that depend
on what lies behind it in memory.  A small C program prints the sizes, the test compares."""
import ctypes
import os
import subprocess

import numpy as np

from mmseqs2_amd import capi
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_STRUCTS = [("mmgpu_sw_params", capi.SwParams), ("mmgpu_sw_query", capi.SwQuery), ("mmgpu_nucl_params", capi.NuclParams),
             ("mmgpu_nucl_query", capi.NuclQuery), ("mmgpu_pf_index", capi.PfIndexDesc), ("mmgpu_pf_params", capi.PfParams),
             ("mmgpu_pf_query", capi.PfQuery), ("mmgpu_pf_shard", capi.PfShard), ("mmgpu_db_info", capi.DbInfo),
             ("mmo_pf_gen", po.PfGen), ("mmo_pf_params", po.PfParams), ("mmo_pf_stats", po.PfStats), ("mmo_pf_profile", po.PfProfile),
             ("mmo_ksw_ez", po.KswEz), ("mmo_nucl_result", po.NuclRes)]
C_RECORDS = [("mmgpu_sw_hit", capi.SW_HIT_DTYPE), ("mmgpu_pf_hit", capi.PF_HIT_DTYPE), ("mmgpu_nucl_pair", capi.NUCL_PAIR_DTYPE),
             ("mmgpu_nucl_hit", capi.NUCL_HIT_DTYPE)]


# (a field that fits into tail padding does not change the size: the last fields are compared by offset as well)
LAST_FIELDS = [("mmgpu_nucl_params", capi.NuclParams, "wrapped"), ("mmgpu_pf_index", capi.PfIndexDesc, "kmer_alphabet"),
               ("mmgpu_pf_params", capi.PfParams, "kmer_score"), ("mmo_pf_params", po.PfParams, "index_base"),
               ("mmgpu_sw_query", capi.SwQuery, None), ("mmgpu_pf_query", capi.PfQuery, None), ("mmgpu_db_info", capi.DbInfo, "file_bytes")]


def test_ctypes_mirrors_have_the_c_sizes(tmp_path):
    names = [n for n, _ in C_STRUCTS + C_RECORDS]
    last = [(n, t, f if f is not None else t._fields_[-1][0]) for n, t, f in LAST_FIELDS]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "mmgpu.h"\n#include "mm_oracle.h"\nint main(void) {\n' + \
          "".join('    printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) + \
          "".join('    printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (n, f, n, f) for n, _, f in last) + "    return 0;\n}\n"
    c = tmp_path / "sizes.c"
    c.write_text(src)
    exe = str(tmp_path / "sizes")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-o", exe, str(c)], check=True)
    out = dict(l.split() for l in subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout.splitlines())
    for n, t in C_STRUCTS:
        assert ctypes.sizeof(t) == int(out[n]), (n, ctypes.sizeof(t), out[n])
    for n, t in C_RECORDS:
        assert np.dtype(t).itemsize == int(out[n]), (n, np.dtype(t).itemsize, out[n])
    for n, t, f in last:
        assert getattr(t, f).offset == int(out[n + "." + f]), (n, f, getattr(t, f).offset, out[n + "." + f])
