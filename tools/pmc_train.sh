#!/bin/bash
# PMC passes (memory counters) over the TRAIN bench command, summarised for the encoder's MSDA backward pair into
# gpurun_out/<tag>_msda_bwd_pmc.json (copy to profiles/ after review; bench.py --mode train attaches it as roofline.traffic).
# Usage: bash tools/pmc_train.sh <tag>
set -u
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/pmc_passes.sh gpurun_out/pmc_train_${tag} train mem -- python3 bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-probes
python3 tools/msda_pmc.py gpurun_out/pmc_train_${tag} --kernel-regex 'msda_bwd_value_tile_f32' --name 'msda_bwd_value_tile_f32' \
    --alg-bytes 359460864 --out gpurun_out/${tag}_msda_bwd_value_pmc.json > gpurun_out/${tag}_msda_bwd_pmc.txt 2>&1
python3 tools/msda_pmc.py gpurun_out/pmc_train_${tag} --kernel-regex 'msda_bwd_q64_f32<false' --name 'msda_bwd_q64_f32<no atomics>' \
    --alg-bytes 359460864 --out gpurun_out/${tag}_msda_bwd_q64_pmc.json >> gpurun_out/${tag}_msda_bwd_pmc.txt 2>&1
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
a = json.load(open(f"gpurun_out/{tag}_msda_bwd_value_pmc.json"))
b = json.load(open(f"gpurun_out/{tag}_msda_bwd_q64_pmc.json"))
def add(k):
    return (a.get(k) or 0) + (b.get(k) or 0) if (a.get(k) is not None and b.get(k) is not None) else None
out = {"kernel": "msda_bwd_q64_f32<no atomics> + msda_bwd_value_tile_f32",
       "source_files": a.get("source_files"), "source_sha256": a.get("source_sha256"),
       "launch": "encoder layer backward, B = 4, Lq = S = 12537 (sum of the two kernels' per-launch averages)",
       "hbm_bytes_per_launch": add("hbm_bytes_per_launch"), "fetch_bytes_corrected": add("fetch_bytes_corrected"), "write_bytes": add("write_bytes"),
       "l1_gather_bytes": add("l1_gather_bytes"), "l2_hit": {"value_tile": a.get("l2_hit"), "q64": b.get("l2_hit")},
       "parts": {"msda_bwd_value_tile_f32": a, "msda_bwd_q64_f32<no atomics>": b}}
json.dump(out, open(f"gpurun_out/{tag}_msda_bwd_pmc.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "fetch_bytes_corrected", "write_bytes", "l2_hit")}))
PY
tail -30 gpurun_out/${tag}_msda_bwd_pmc.txt
find gpurun_out/pmc_train_${tag} -name "*.db" -delete
