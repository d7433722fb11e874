for v in 15 14; do for j in 0 0.5 1 2 4; do python tools/msda_bench.py --variant $v --jitter $j --graph --fused 2>&1 | tail -1 | cut -c1-80; done; done
