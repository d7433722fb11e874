cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_fb
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-kernel-probes --extras 0 > /dev/null 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 60 seq > gpurun_out/r06_forward_sequence.txt 2>&1
rm -rf gpurun_out/prof_fb
wc -l gpurun_out/r06_forward_sequence.txt
