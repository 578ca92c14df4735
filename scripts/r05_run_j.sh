O=gpurun_out/r05j; mkdir -p $O
python -m pytest tests/test_db_file.py -x -q > $O/test_db.log 2>&1; tail -3 $O/test_db.log
hl() { python bench.py --headline-only --no-cpu-baseline --steps 5 2>$O/err_$1.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_stages'], d['prefilter']['stage_ms'])"; }
{
hl base
MMGPU_LIB=$PWD/variants/maxr32/libmmgpu.so hl maxr32
MMGPU_LIB=$PWD/variants/maxr24/libmmgpu.so hl maxr24
} > $O/sw_variants.txt 2>&1
cat $O/sw_variants.txt
python scripts/exp_nucl_search.py 50000 1000 2>$O/nucl.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('setup_s', 'prefilter_ms', 'align_ms', 'strand_queries', 'hits')})" > $O/nucl.txt 2>&1
cat $O/nucl.txt
