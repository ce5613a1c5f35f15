for nb in 512 1024; do echo "== RGBD360_EVAL_BLOCKS=$nb"; RGBD360_EVAL_BLOCKS=$nb python tools/quick_perf.py | grep -E "hg 1|forced|full align|level 3"; done
