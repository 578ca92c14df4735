O=gpurun_out/r06s; mkdir -p $O
df -h /dev/shm /tmp | head -5; mount | grep -E " /tmp | / |/dev/shm" | head -4
TMPDIR=/dev/shm python scripts/search_timeline.py 10 > $O/search_timeline_shm.txt 2>&1; head -2 $O/search_timeline_shm.txt; grep -n "TRACEX\|Calculation of\|open device\|Time for processing\|10.00K\|merging to res" $O/search_timeline_shm.txt | cut -c1-120
