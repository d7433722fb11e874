for a in 0 1 2 3 7; do echo "ablate=$a"; EGTR_REGION_ABLATE=$a python tools/msda_bench.py --variant 15 --jitter 0 --graph --fused 2>&1 | tail -1 | cut -c1-80; done
for v in 14 15; do for j in 0 0.5 1 2 4; do python tools/msda_bench.py --variant $v --jitter $j --graph --fused 2>&1 | tail -1 | cut -c1-80; done; done
