O=gpurun_out/r05q; mkdir -p $O
python -m pytest tests/test_profile_query.py tests/test_sw_gpu.py -x -q -m gpu -k "block or profile" > $O/test_block.log 2>&1; tail -3 $O/test_block.log
python -m pytest tests/test_mmseqs_dropin.py -x -q -m gpu -k "profile" > $O/test_dropin_profile.log 2>&1; tail -3 $O/test_dropin_profile.log
