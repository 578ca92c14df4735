O=gpurun_out/r06c; mkdir -p $O
python bench.py --no-cpu-baseline --headline-only --steps 10 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
