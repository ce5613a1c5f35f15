for nb in 64 128 256 512; do echo "== 1024-thread blocks, RGBD360_EVAL_BLOCKS=$nb"; RGBD360_EVAL_BLOCKS=$nb python tools/quick_perf.py | grep -E "hg 1|forced|full"; done
