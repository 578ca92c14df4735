O=gpurun_out/r05k; mkdir -p $O
sleep 2
python -m pytest tests/test_db_file.py tests/test_mmseqs_dropin.py -x -q -k "db or persisted" > $O/test_db.log 2>&1; tail -3 $O/test_db.log
python scripts/exp_nucl_search.py 50000 1000 2>$O/nucl.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: v for k, v in d.items() if not isinstance(v, (dict, list)) or k in ('setup_s',)}))" > $O/nucl.txt 2>&1
cat $O/nucl.txt | cut -c1-1500
