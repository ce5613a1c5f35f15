for pc in 1 2 3 4 6; do echo "== RGBD360_POLL_CHUNK=$pc"; RGBD360_POLL_CHUNK=$pc python tools/quick_perf.py | grep -E "full align"; done
