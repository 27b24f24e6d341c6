# Development tool: timing of the Hamming scan for several lengths of its insertion phase (tiles of 32 rows).
set -e
for s in ${SWEEP:-8 16 24 32 40 48}; do
  echo "== sort_tiles $s"
  ORBHIP_KNN_SORT_TILES=$s timeout -k 10 120 python tools/knn_bench.py 2>&1 | grep -v amdgpu.ids
done
if [ -n "$TESTS" ]; then timeout -k 10 300 python -u -m pytest tests/test_gpu_matcher.py tests/test_gpu_sharded.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread; fi
