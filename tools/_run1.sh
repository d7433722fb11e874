cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/st -o st -- python tools/stress_bench.py --iters 4 > gpurun_out/st.log 2>&1
python tools/rocpd_stats.py gpurun_out/st/st_results.db --last-ms 200 --top 40 > gpurun_out/stress_stats.txt 2>&1
rm -rf gpurun_out/st
head -45 gpurun_out/stress_stats.txt | cut -c1-165
