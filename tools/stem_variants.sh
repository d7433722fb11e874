cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/stem_x6.o")
for v in 3 4 2 1; do
  d=/tmp/sv_$v; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc -DEGTR_STEM_PH=$v -c egtr_amd/csrc/stem_x6.hip -o $d/t.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/t.o -o $d/lib.so || continue
  echo "=== pooled rows per workgroup $v"
  EGTR_HIP_LIBRARY=$d/lib.so timeout 300 python3 tools/stem_ab.py 2>&1 | grep "one launch"
done
