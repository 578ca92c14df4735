O=gpurun_out/r06r; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_err.txt; tail -c 300 $O/bench_n1.json; echo
python scripts/search_timeline.py 10 > $O/search_timeline.txt 2>&1; head -3 $O/search_timeline.txt
