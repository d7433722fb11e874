#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command on the final round-6 tree -> profiles/r06_bench_kernel_stats.txt
# (the per-kernel averages the bench line's live-timed roofline entries must agree with) and the traced run's own JSON line.
set -u
tag=r06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python3 bench.py > gpurun_out/${tag}_bench_rocprofv3.log 2>&1
grep -E '^\{"metric"' gpurun_out/${tag}_bench_rocprofv3.log > gpurun_out/${tag}_bench_under_rocprofv3.json
{
echo "# rocprofv3 --kernel-trace --stats over the default command (python3 bench.py), final round-6 tree (backbone kernels of DESIGN 4.7b-d in)."
echo "# The default command launches msda_fwd_q64_f32 at several sizes (headline workload: 3135 workgroups = 12 537 queries; train leg: B = 4;"
echo "# mixed_shapes leg: eleven other image shapes): the line 'exactly 3135 workgroups' is the kernel bench.py's roofline entry times live."
python3 tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db --top 70 --split-grid msda_fwd_q64:1000 --exact-grid msda_fwd_q64:3135
} > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_bench
cut -c1-300 gpurun_out/${tag}_bench_under_rocprofv3.json
grep -E "exactly|conv3x3|conv_tail|stem_x6|ffn_x6|rel_head_fwd|decoder_layer" gpurun_out/${tag}_bench_kernel_stats.txt | cut -c1-170
