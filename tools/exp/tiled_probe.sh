R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
echo "== row-major"; N=999936 QS=1,32,100,128 python3 $R/tools/bench_search.py 2>&1 | grep "^Q="
echo "== tiled addressing (timing probe, garbage results)"; LRX_LIB_DEV_VARIANT=$R/lightretriever_amd/build/liblrx_tp.so N=999936 QS=1,32,100,128 python3 $R/tools/bench_search.py 2>&1 | grep "^Q="
done
